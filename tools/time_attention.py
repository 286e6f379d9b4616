"""Attention core per call (B = 32, H = 6): attn_flash16.hip (L <= 96) / attn_long.hip (L <= 160) -- graph replay, HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
from bench import time_kernel
from upp_hip import ops
B, H = 32, 6
for L in (27, 32, 35, 64, 65, 75, 96, 128, 129, 139):
    qkv = torch.randn(B, L, 3 * H * 64, device='cuda')
    ctx, lse = ops.attn_fwd(qkv, B, L, H, 0.125)
    tf = time_kernel(lambda: ops.attn_fwd(qkv, B, L, H, 0.125))
    tb = time_kernel(lambda: ops.attn_bwd(qkv, ctx, ctx, lse, B, L, H, 0.125))
    print("L=%d  fwd %.1f us bwd %.1f us" % (L, tf * 1e3, tb * 1e3), flush=True)
