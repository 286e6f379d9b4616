"""Every upp_linear_f32 / upp_linear_sb_f32 launch of one eager step of a workload, grouped by (M, N, K, epilogue): count, tile code, mean microseconds,
TFLOP/s.    python tools/linear_calls.py [headline|seg|stage2|pretrain|pretask|cls_aux]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from upp_hip import _abi, ops  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "seg"
    dev = torch.device("cuda", 0)
    if kind == "headline":
        ts = bench.Trainer(dev, 32, False, use_graph=False, pipeline=False).ts
    else:
        ts = bench.RecipeTrainer(kind, dev, 32, use_graph=False).ts
    for _ in range(3):
        ts._forward_backward()
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for _ in range(5):
        with ops.time_linear_calls() as scope:
            ts._forward_backward()
        for M, N, K, e, ms, sb in scope.report():
            a = agg.setdefault((M, N, K, e, sb), [0, 0.0])
            a[0] += 1; a[1] += ms
    lib = _abi.load()
    total = 0.0
    for (M, N, K, e, sb), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        us = ms * 1e3 / n
        total += ms * 1e3 / 5
        print("%6d x %5d x %5d epi %d  x%-3d tile %8x  %7.1f us  %5.1f TF  %7.1f us/step" % (M, N, K, e, n // 5, sb or lib.upp_linear_tile(M, N, K), us, 2.0 * M * N * K / us / 1e6, ms * 1e3 / 5))
    print("total %.1f us/step (eager, event-bracketed)" % total)


if __name__ == "__main__":
    main()
