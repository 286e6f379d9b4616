"""The grouped weight-gradient launches of one step of a workload, re-timed stand-alone on random operands: exact f32 against split bf16.
    python tools/micro/wgrad_groups.py [seg|pretrain|pretask|stage2|cls_aux]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
import bench
from upp_hip import ops

kind = sys.argv[1] if len(sys.argv) > 1 else "seg"
ts = bench.RecipeTrainer(kind, torch.device("cuda", 0), 32, use_graph=False).ts
for _ in range(2):
    ts._forward_backward()
with ops.time_linear_calls() as scope:
    ts._forward_backward()
torch.cuda.synchronize()
groups = list(scope.wgrad_groups)
for grp in list(groups):
    if len(grp) > 8:         # the large group again, by tile class
        groups.append([s for s in grp if s[1] > 128 and s[2] > 128])
        groups.append([s for s in grp if not (s[1] > 128 and s[2] > 128)])
for grp in groups:
    pairs = [(torch.randn(M, N, device='cuda'), torch.randn(M, K, device='cuda')) for M, N, K in grp]
    gf = sum(2.0 * M * N * K for M, N, K in grp) / 1e9
    line = "%3d problems %8.1f GF" % (len(grp), gf)
    for split in (False, True):
        ops.WGRAD_SPLIT_BF16 = split
        for _ in range(2):
            ops.linear_wgrad_grouped(pairs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            parts = ops.linear_wgrad_grouped(pairs)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 200
        line += "   %s %8.1f us %6.1f TF  partials %.0f MB" % ("sb " if split else "f32", us, gf * 1e3 / us, sum(p.numel() for p in parts) * 4 / 1e6)
    print(line)
    shapes = {}
    for s in grp:
        shapes[s] = shapes.get(s, 0) + 1
    print("      ", ", ".join("%dx %s" % (n, s) for s, n in shapes.items()))
