import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'iccv2025-upp_amd')
import torch, bench
from utils import synthetic as _seeded
from upp_hip import ops
a = _seeded.unit_ball_clouds(32, 1024, 1).cuda(); b = _seeded.unit_ball_clouds(32, 1024, 2).cuda()
t = bench.time_kernel(lambda: ops.emd_approxmatch(a, b), iters=3)
print("emd approxmatch (32,1024)x(1024): %.1f us" % (t * 1e3))
m = ops.emd_approxmatch(a, b)
t = bench.time_kernel(lambda: ops.emd_matchcost(a, b, m), iters=5)
print("emd matchcost: %.1f us" % (t * 1e3))
