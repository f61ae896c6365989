"""e2ehip -- host-side plumbing for libe2eslam_hip.so (MI355X / gfx950 only)."""
from . import _lib
from ._lib import E2EError, load
from . import ops

__all__ = ["_lib", "ops", "load", "E2EError"]
