"""upp_hip -- MI355X (gfx950) operators for the UPP / Point-MAE hot path.

Layout:
  csrc/        hand-written HIP kernels + the C ABI of include/upp_hip.h
  build.py     hipcc driver (in-tree libupp_hip.so)
  _abi.py      ctypes binding (no fallback: missing library -> RuntimeError)
  ops.py       tensor validation + launches on torch's current stream
  functional.py  autograd Functions with the reference operators' semantics
The packages next to this one (pointnet2_ops, knn_cuda, extensions, emd, chamfer,
emd_cuda, utils, models) mirror the import surface of the reference so that
`Point_MAE_unify` code written against it runs unchanged.
"""
from . import _abi  # noqa: F401
from .functional import (  # noqa: F401
    furthest_point_sample, gather_operation, fps_gather, knn_query, knn_group, group_points,
    ChamferFunction, EarthMoverDistanceFunction, rowln, attention, prop_pool, prop_interp, adapter, PropIndex, propagate,
)

__all__ = ["furthest_point_sample", "gather_operation", "fps_gather", "knn_query", "knn_group",
           "group_points", "ChamferFunction", "EarthMoverDistanceFunction"]
