"""The one name of `kornia` the reference imports (online_adaption.py:15, imported and never called there)."""
from . import geometry  # noqa: F401
