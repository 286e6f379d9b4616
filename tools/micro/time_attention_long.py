import os, sys
ROOT='/root/repo'
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
from bench import time_kernel
from upp_hip import ops
B, H = 32, 6
for L in (96, 97, 112, 128, 129, 130, 132, 133, 138, 139, 144, 160):
    qkv = torch.randn(B, L, 3 * H * 64, device='cuda')
    ctx, lse = ops.attn_fwd(qkv, B, L, H, 0.125)
    tf = time_kernel(lambda: ops.attn_fwd(qkv, B, L, H, 0.125))
    tb = time_kernel(lambda: ops.attn_bwd(qkv, ctx, ctx, lse, B, L, H, 0.125))
    fl = 4.0 * B * H * L * L * 64
    print("L=%d fwd %.1f us (%.0f TF) bwd %.1f us (%.0f TF)" % (L, tf * 1e3, fl / tf / 1e9, tb * 1e3, 2.5 * fl / tb / 1e9), flush=True)
