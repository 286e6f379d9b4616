"""Kernel-level timing sweeps on the GPU box (tuning aid; prints a table).
    python tools/sweep_kernels.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "tests", "iccv2025-upp_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402

from utils import synthetic as _seeded  # noqa: E402
sys.path.insert(0, ROOT)
from bench import time_kernel  # noqa: E402
from upp_hip import _abi, ops  # noqa: E402


def main():
    lib = _abi.load()
    B = 32
    print("FPS  (B=%d)  ms per call by waves/cloud" % B)
    for (N, M) in [(1024, 64), (1096, 32), (972, 32), (1024, 256), (1228, 1024), (64, 32), (32, 32), (6144, 1024)]:
        x = _seeded.unit_ball_clouds(B, N, seed=N).cuda()
        row = []
        for w in (1, 2, 4, 8):
            row.append(time_kernel(lambda: ops.fps(x, M, want_centers=True, waves=w), iters=10, warm=2))
        row.append(time_kernel(lambda: ops.fps(x, M, want_centers=True), iters=10, warm=2))
        print("  N=%5d M=%5d  W1 %.4f  W2 %.4f  W4 %.4f  W8 %.4f  auto %.4f   us/iter(best) %.3f"
              % (N, M, *row, 1000 * min(row[:4]) / max(M - 1, 1)))
    print("kNN (B=%d) ms per call, prefilter off / on" % B)
    for (N, Q, K) in [(1024, 64, 32), (1096, 32, 16), (972, 32, 16), (64, 32, 8), (32, 32, 16), (1536, 128, 32), (8192, 64, 32)]:
        x = _seeded.unit_ball_clouds(B, N, seed=N).cuda()
        q = x[:, :Q].contiguous()
        row = []
        for on in (0, 1):
            row.append(time_kernel(lambda: ops.knn(x, q, K, want_dist=False, want_neigh=True, prefilter=on), iters=10, warm=2))
        print("  N=%5d Q=%4d K=%3d  off %.4f  on %.4f" % (N, Q, K, *row))
    a = _seeded.unit_ball_clouds(B, 1024, seed=1).cuda()
    b = _seeded.unit_ball_clouds(B, 1024, seed=2).cuda()
    t = time_kernel(lambda: ops.chamfer_fwd(a, b), iters=10)
    d1, d2, i1, i2 = ops.chamfer_fwd(a, b)
    tb = time_kernel(lambda: ops.chamfer_bwd(a, b, i1, i2, d1, d2), iters=10)
    print("chamfer fwd (32,1024,1024) %.4f ms   bwd %.4f ms" % (t, tb))
    t = time_kernel(lambda: ops.emd_approxmatch(a, b), iters=3, warm=1)
    m = ops.emd_approxmatch(a, b)
    tc = time_kernel(lambda: ops.emd_matchcost(a, b, m), iters=3, warm=1)
    g = torch.ones(B, device='cuda')
    tg = time_kernel(lambda: ops.emd_matchcost_bwd(g, a, b, m), iters=3, warm=1)
    print("emd approxmatch (32,1024,1024) %.3f ms  matchcost %.3f ms  matchcost_bwd %.3f ms" % (t, tc, tg))


if __name__ == "__main__":
    main()
