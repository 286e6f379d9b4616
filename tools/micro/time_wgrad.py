"""Weight-gradient kernels on the segmentation head's shapes: exact f32 (linear_rt.hip) against split bf16 (wgrad_sb.hip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "iccv2025-upp_amd"))
import torch
from upp_hip import ops

SHAPES = [(65536, 1536, 1024), (65536, 1024, 512), (65536, 512, 256), (65536, 256, 52), (4128, 384, 1536), (4128, 1536, 384), (4128, 1152, 384), (2400, 384, 384)]
for M, N, K in SHAPES:
    g, x = torch.randn(M, N, device='cuda'), torch.randn(M, K, device='cuda')
    line = "%6d x %5d x %5d" % (M, N, K)
    for split in (False, True):
        ops.WGRAD_SPLIT_BF16 = split
        for _ in range(3):
            ops.linear_wgrad_grouped([(g, x)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            p = ops.linear_wgrad_grouped([(g, x)])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        line += "   %s %8.1f us %6.1f TF (splits %d)" % ("sb " if split else "f32", us, 2.0 * M * N * K / us / 1e6, p[0].shape[0])
    print(line)
