"""Call every stand-alone operator a few times (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "tests", "iccv2025-upp_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402

from utils import synthetic as _seeded  # noqa: E402
from upp_hip import ops  # noqa: E402

B = 32
x = _seeded.unit_ball_clouds(B, 1024, seed=1).cuda()
y = _seeded.unit_ball_clouds(B, 1024, seed=2).cuda()
x1228 = _seeded.unit_ball_clouds(B, 1228, seed=3).cuda()
for _ in range(10):
    idx, cen = ops.fps(x, 64, want_centers=True)
    ops.knn(x, cen, 32, want_dist=False, want_neigh=True)
    ops.fps(x1228, 1024, want_centers=True)
    d1, d2, i1, i2 = ops.chamfer_fwd(x, y)
    ops.chamfer_bwd(x, y, i1, i2, d1, d2)
for _ in range(3):
    m = ops.emd_approxmatch(x, y)
    c = ops.emd_matchcost(x, y, m)
    ops.emd_matchcost_bwd(torch.ones(B, device='cuda'), x, y, m)
torch.cuda.synchronize()
print("done")
