from .icpslam import ICPSLAM  # noqa: F401
from .pointfusion import PointFusion  # noqa: F401
from . import fusionutils  # noqa: F401
