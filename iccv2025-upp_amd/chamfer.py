"""Drop-in for the compiled `chamfer` module (reference extensions/chamfer_dist/chamfer_cuda.cpp:36-39)."""
from upp_hip import ops


def forward(xyz1, xyz2):
    return list(ops.chamfer_fwd(xyz1, xyz2))


def backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2):
    return list(ops.chamfer_bwd(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2))
