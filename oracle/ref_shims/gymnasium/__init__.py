"""Minimal stand-in for `gymnasium`, used ONLY to import the reference here.

TEST INFRASTRUCTURE (oracle side).  The reference's task layer builds
`gymnasium.spaces.Box / Dict` objects to DESCRIBE its action and observation
spaces (reference ratinabox/contribs/TaskEnvironment.py:15, 117-118, 197-203);
they take no part in stepping.  gymnasium is not installed in this image.
"""
from . import spaces  # noqa: F401
