"""Attention core per call (B = 32, H = 6): variants 0 (attn_flash16 / attn_long), 2 (attn_mfma, round 1), 1 (VALU) -- graph replay, HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
from bench import time_kernel
from upp_hip import ops
B, H = 32, 6
for L in (27, 32, 35, 64, 65, 75, 96):
    qkv = torch.randn(B, L, 3 * H * 64, device='cuda')
    row = ["L=%d" % L]
    for v in (0, 2, 1):
        ctx, lse = ops.attn_fwd(qkv, B, L, H, 0.125, v)
        tf = time_kernel(lambda: ops.attn_fwd(qkv, B, L, H, 0.125, v))
        tb = time_kernel(lambda: ops.attn_bwd(qkv, ctx, ctx, lse, B, L, H, 0.125, v))
        row.append("v%d fwd %.1f us bwd %.1f us" % (v, tf * 1e3, tb * 1e3))
    print("  ".join(row), flush=True)
