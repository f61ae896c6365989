from .pointclouds import Pointclouds  # noqa: F401
from .rgbdimages import RGBDImages  # noqa: F401
