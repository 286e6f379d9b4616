import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'iccv2025-upp_amd')
import torch, bench
from upp_hip import ops, functional as HF
B = 32
lp, cen = torch.rand(B, 2048, 3, device='cuda'), torch.rand(B, 128, 3, device='cuda')
z = torch.randn(B, 128, 1536, device='cuda')
wt = torch.randn(3, 1536, device='cuda')
d, i = HF.sqdist_topk(lp, cen, 3)
g = torch.randn(B, 2048, 1536, device='cuda')
t = bench.time_kernel(lambda: ops.interp_affine_fwd(d, i, z, lp, wt, 3, 1e-4), iters=5)
print("interp_wide fwd (32,2048)x1536: %.1f us (%.2f TB/s written)" % (t * 1e3, g.numel() * 4 / t / 1e9))
t = bench.time_kernel(lambda: ops.interp_bwd(d, i, g, 128, 3, 1e-4), iters=5)
print("interp_bwd: %.1f us" % (t * 1e3))
