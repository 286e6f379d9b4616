"""Drop-in for the `pointnet2_ops` package the reference imports (utils/misc.py:10)."""
from . import pointnet2_utils  # noqa: F401

__version__ = "3.0.0+upp_hip"
