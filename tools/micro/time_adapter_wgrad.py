#!/usr/bin/env python3
"""upp_adapter_wgrad_batched stand-alone: 12 blocks of R rows (graph replays, HIP events).   python tools/micro/time_adapter_wgrad.py [R ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402

from bench import time_kernel  # noqa: E402
from upp_hip import ops  # noqa: E402

D, H, dev = 384, 32, "cuda"
for R in [int(a) for a in sys.argv[1:]] or [2080, 2400, 4128]:
    jobs = []
    for _ in range(12):
        jobs.append((torch.randn(R, D, device=dev), torch.randn(R, device=dev), torch.rand(R, device=dev) + 0.5, torch.randn(D, device=dev),
                     torch.randn(D, device=dev), torch.randn(R, D, device=dev), torch.randn(R, 2 * H, device=dev), 0.7))
    ms = time_kernel(lambda: ops.adapter_wgrad_batched(jobs))
    mb = 12 * R * (2 * D + 2 * H + 2) * 4 / 1e6
    parts = ops.adapter_wgrad_batched(jobs)
    print("R = %5d   %6.1f us   %5.1f MB read -> %.2f TB/s   splits %d" % (R, ms * 1e3, mb, mb / ms / 1e3 / 1e3, parts[0].shape[0]), flush=True)
