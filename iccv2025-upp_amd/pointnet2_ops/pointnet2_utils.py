"""pointnet2_ops.pointnet2_utils surface used by UPP (reference utils/misc.py:18-19,
tools/runner_module.py:151-153), served by the gfx950 kernels of libupp_hip.so."""
from upp_hip.functional import (  # noqa: F401
    FurthestPointSampling, GatherOperation, furthest_point_sample, gather_operation,
)

__all__ = ["FurthestPointSampling", "GatherOperation", "furthest_point_sample", "gather_operation"]
