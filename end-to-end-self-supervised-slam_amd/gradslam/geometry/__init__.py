from . import geometryutils  # noqa: F401
