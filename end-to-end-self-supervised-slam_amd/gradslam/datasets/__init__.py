"""ICL-NUIM and TUM RGB-D sequence loaders with gradslam.datasets' item contract.
Host-side file formats only (PNG decoding through Pillow); the frame stack they return is uploaded to HBM once by the
driver and stays resident (online_adaption.py:211-220)."""
from .icl import ICL  # noqa: F401
from .tum import TUM  # noqa: F401
