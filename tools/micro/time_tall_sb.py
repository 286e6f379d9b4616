import sys, os
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
from bench import time_kernel
from upp_hip import ops
for M, N, K in ((65536, 512, 256), (65536, 384, 512), (16384, 512, 256), (16384, 384, 512), (65536, 256, 128), (65536, 1024, 1536)):
    a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * K ** -0.5; w._upp_persistent = True
    out = torch.empty(M, N, device='cuda')
    ops.linear_f32(a, w, out=out, frozen=True)
    tsb = time_kernel(lambda: ops.linear_f32(a, w, out=out, frozen=True), iters=10)
    t32 = time_kernel(lambda: ops.linear_f32(a, w, out=out), iters=10)
    fl = 2.0 * M * N * K
    print("%6d x %4d x %4d  sb %7.1f us (%5.1f TF, tile %x)   f32 %7.1f us (%5.1f TF)" % (M, N, K, tsb * 1e3, fl / tsb / 1e9, ops.linear_sb_tile(M, N, K), t32 * 1e3, fl / t32 / 1e9), flush=True)
