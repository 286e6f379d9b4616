"""Tall split-bf16 Linear shapes (segmentation head, patch embedding) per tile:  python tools/micro/time_tall_tiles.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
from bench import time_kernel
from upp_hip import ops, _abi
TILES = [0x484412, 0x484212, 0x444213, 0x443114, 0x434214]
dev = torch.device("cuda", 0)
for M, N, K in ((65536, 1024, 1536), (65536, 1536, 1024), (65536, 512, 1024), (65536, 1024, 512), (65536, 512, 256), (65536, 384, 512), (16384, 512, 256), (16384, 384, 512), (4128, 1536, 384), (4128, 384, 1536)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; w._upp_persistent = True
    out = torch.empty(M, N, device=dev)
    ref = ops.linear_f32(a, w)
    planes = ops.PLANES.get(w)
    line = "%6d x %4d x %4d  pick %x " % (M, N, K, ops.linear_sb_tile(M, N, K))
    for t in TILES:
        def run(t=t):
            ops._call(dev, "upp_linear_sb_f32", _abi.ptr(a), K, _abi.ptr(planes), None, _abi.ptr(out), N, None, 0, M, N, K, 0, t)
        run()
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        us = time_kernel(run, iters=10) * 1e3
        line += "  %x: %7.1f us %5.1f TF (%.0e)" % (t, us, 2.0 * M * N * K / us / 1e6, err)
    print(line, flush=True)
