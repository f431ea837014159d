from . import pointnet2_utils  # noqa: F401
