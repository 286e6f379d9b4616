import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'iccv2025-upp_amd')
import torch, bench
from utils import synthetic as _seeded
from upp_hip import ops
for B, n, m in ((32, 1024, 1024), (32, 2048, 8192), (1216, 32, 32)):
    a = _seeded.unit_ball_clouds(B, n, 1).cuda(); b = _seeded.unit_ball_clouds(B, m, 2).cuda()
    t = bench.time_kernel(lambda: ops.chamfer_fwd(a, b))
    print("chamfer fwd (%d,%d)x(%d): %.1f us (both directions)" % (B, n, m, t * 1e3))
