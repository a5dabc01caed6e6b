"""ratinabox_amd — MI355X-native batched drop-in for RatInABox's per-step hot path.

`Environment`, `Agent`, `PlaceCells`, `GridCells`, `BoundaryVectorCells`,
`HeadDirectionCells` keep the reference's Python surface; `Agent.update()` and
`Neurons.update()/get_state()` run as hand-written HIP kernels for gfx950 behind
the C ABI in include/riab_hip.h.  Importing this package loads (building it if
needed) libriab_hip.so and raises if that is impossible: there is no CPU path."""
verbose = False

from . import _lib  # noqa: E402,F401  (fails loudly when the HIP library is unavailable)
from . import utils  # noqa: E402,F401
from .Environment import Environment  # noqa: E402,F401
from .Agent import Agent  # noqa: E402,F401
from .Neurons import (  # noqa: E402,F401
    Neurons, PlaceCells, GridCells, VectorCells, BoundaryVectorCells, FieldOfViewBVCs, ObjectVectorCells,
    FieldOfViewOVCs, AgentVectorCells, FieldOfViewAVCs, HeadDirectionCells, VelocityCells, SpeedCell,
    RandomSpatialNeurons, FeedForwardLayer)

__all__ = ["Environment", "Agent", "Neurons", "PlaceCells", "GridCells", "VectorCells", "BoundaryVectorCells",
           "FieldOfViewBVCs", "ObjectVectorCells", "FieldOfViewOVCs", "AgentVectorCells", "FieldOfViewAVCs",
           "HeadDirectionCells", "VelocityCells", "SpeedCell", "RandomSpatialNeurons", "FeedForwardLayer", "utils"]
