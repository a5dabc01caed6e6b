"""ratinabox_amd — MI355X-native batched drop-in for RatInABox's per-step hot path.

`Environment`, `Agent`, `PlaceCells`, `GridCells`, `BoundaryVectorCells`,
`HeadDirectionCells` keep the reference's Python surface; `Agent.update()` and
`Neurons.update()/get_state()` run as hand-written HIP kernels for gfx950 behind
the C ABI in include/riab_hip.h.  Importing this package loads (building it if
needed) libriab_hip.so and raises if that is impossible: there is no CPU path."""
verbose = False

import os as _os

# simulate() runs two kernels side by side on two streams.  The HIP runtime multiplexes a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues (4 by default); in a process that also holds an RCCL communicator (torch.distributed,
# backend "nccl") that was too few: the trajectory kernel's stream shared a queue with the caller's and the two kernels
# ran one after the other (cfg 2, 20 steps: 131 instead of 90 us per call; 90 again with 8 queues [MI355X]).  The runtime
# reads the variable when it initialises (the first HIP call of the process), so it is set here — unless the user has
# set it, or has asked for the process environment to be left alone (RIAB_KEEP_HW_QUEUES=1: child processes inherit
# os.environ) — with a warning if the runtime was already up.  Round 4 added a second line of defence that needs no
# environment variable: the library TIMES candidate streams against the caller's stream and sets aside those that
# share its hardware queue (csrc/riab_simulate.hip side_stream_for, Agent.pipeline_info()); with more queues there
# are more good candidates.
_hwq_preset = "GPU_MAX_HW_QUEUES" in _os.environ or _os.environ.get("RIAB_KEEP_HW_QUEUES") == "1"
if not _hwq_preset:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"

from . import _lib  # noqa: E402,F401  (fails loudly when the HIP library is unavailable)

if not _hwq_preset:
    import torch as _torch
    if _torch.cuda.is_initialized():
        import warnings as _warnings
        _warnings.warn("ratinabox_amd was imported after the HIP runtime had been initialised with its default of 4 hardware "
                       "queues: in a process that also uses torch.distributed (RCCL), Agent.simulate() may run its two "
                       "kernels one after the other.  Set GPU_MAX_HW_QUEUES=8 in the environment, or import ratinabox_amd "
                       "before the first CUDA call.")
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
