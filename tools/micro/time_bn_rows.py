"""BatchNorm-over-rows kernels at the per-point heads' shapes: microseconds and bytes / s per launch (HIP events, many launches).
   python tools/micro/time_bn_rows.py [launches] [R,C ...]      (shapes: default the heads' list; one shape under rocprofv3 --stats
   gives the per-kernel split)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402

from upp_hip import ops  # noqa: E402


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = "cuda"
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(65536, 1536), (65536, 1024), (65536, 512), (65536, 256), (65536, 128),
                                                                            (16384, 512), (4096, 384), (2048, 512)]
    for R, C in shapes:
        x = torch.randn(R, C, device=dev)
        g = torch.randn(R, C, device=dev)
        ga, be = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y, mean, rstd = ops.bn_rows_fwd(x, ga, be, rm, rv, 0.1, 1e-5, True, True, want_stats=True)
        mb = R * C * 4 / 1e6
        t_f = timed(lambda: ops.bn_rows_fwd(x, ga, be, rm, rv, 0.1, 1e-5, True, True, want_stats=True), n)
        t_e = timed(lambda: ops.bn_rows_fwd(x, ga, be, rm, rv, 0.1, 1e-5, False, True), n)
        t_b = timed(lambda: ops.bn_rows_bwd(x, g, mean, rstd, ga, be, True), n)
        t_s = timed(lambda: ops.bn_rows_bwd(x, g, mean, rstd, ga, be, True, want_gx=False), n)
        print("%6d x %4d (%6.1f MB)  fwd train %7.1f us (3 passes: %.2f TB/s)  apply alone %7.1f us (%.2f TB/s)  bwd %7.1f us (5 passes: %.2f TB/s)"
              "  bwd apply alone %7.1f us (%.2f TB/s)" % (R, C, mb, t_f, 3 * mb / t_f, t_e, 2 * mb / t_e, t_b, 5 * mb / t_b, t_b - t_s, 3 * mb / max(t_b - t_s, 1e-3)))


if __name__ == "__main__":
    main()
