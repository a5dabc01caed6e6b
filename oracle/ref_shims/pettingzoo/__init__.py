"""Minimal stand-in for `pettingzoo`, used ONLY to import the reference here.

TEST INFRASTRUCTURE (oracle side).  The reference's task layer subclasses
`pettingzoo.ParallelEnv` (reference ratinabox/contribs/TaskEnvironment.py:14, 30)
purely as an interface marker: none of its methods are inherited on the step
path (step / reset / observation_space / action_space are all overridden).
pettingzoo is not installed in this image and cannot be installed (no network).
Never imported by the product package; never travels to the GPU box as a code
path (tests/golden/task_*.npz are data generated with it).
"""


class ParallelEnv:
    """Interface marker only."""
    pass
