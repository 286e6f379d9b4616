"""Drop-in for the compiled `emd_cuda` module (reference extensions/emd/cuda/emd.cpp:23-27)."""
from upp_hip import ops


def approxmatch_forward(xyz1, xyz2):
    return ops.emd_approxmatch(xyz1, xyz2)


def matchcost_forward(xyz1, xyz2, match):
    return ops.emd_matchcost(xyz1, xyz2, match)


def matchcost_backward(grad_cost, xyz1, xyz2, match):
    return list(ops.emd_matchcost_bwd(grad_cost, xyz1, xyz2, match))
