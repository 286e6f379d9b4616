"""A/B timing of csrc/linear.hip built with extra -D macros (diagnostic / experimental switches), without touching the product
library:   python tools/micro/lin_ab.py [MACRO ...] [-- shape:tile ...]
Prints device time per call (HIP-graph replay, bench.time_kernel) of every Transformer-block shape at the default tile, or
at the tiles given as name:hex."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402
from upp_hip.build import FLAGS as PRODUCT_FLAGS  # noqa: E402  (the flag list of the shipped library, -packed-fp32-ops included: a micro build measures the same code)

from bench import time_kernel  # noqa: E402

SRC = os.path.join(ROOT, "iccv2025-upp_amd", "upp_hip", "csrc", "linear.hip")
SHAPES = {"qkv_2400": (2400, 1152, 384), "proj_2400": (2400, 384, 384), "fc1_2400": (2400, 1536, 384), "fc2_2400": (2400, 384, 1536),
          "dqkv_2400": (2400, 384, 1152), "fc1_2048": (2048, 1536, 384), "qkv_1120": (1120, 1152, 384), "fc1_1120": (1120, 1536, 384),
          "fc2_1120": (1120, 384, 1536), "proj_1120": (1120, 384, 384)}


def main():
    argv = sys.argv[1:]
    picks = []
    if "--" in argv:
        i = argv.index("--")
        argv, picks = argv[:i], argv[i + 1:]
    so = "/tmp/liblin_ab_%s.so" % ("_".join(argv) or "base")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + PRODUCT_FLAGS + ["-shared"]
                          + ["-D" + a for a in argv] + [SRC, os.path.join(os.path.dirname(SRC), "linear_rt.hip"), os.path.join(os.path.dirname(SRC), "abi.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    vp, ll, ci = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
    lib.upp_linear_f32.argtypes = [vp, ll, vp, ll, vp, vp, ll, vp, ll, ci, ci, ci, ci, ci, vp]
    dev = torch.device("cuda", 0)
    jobs = [(n, 0) for n in SHAPES] if not picks else [(p.split(":")[0], int(p.split(":")[1], 16)) for p in picks]
    out = []
    for name, tile in jobs:
        M, N, K = SHAPES[name]
        a = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        c = torch.empty(M, N, device=dev)
        ref = a @ w.t()

        def call():
            rc = lib.upp_linear_f32(a.data_ptr(), K, w.data_ptr(), K, None, c.data_ptr(), N, None, 0, M, N, K, 0, tile,
                                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        call()
        err = ((c - ref).abs().max() / ref.abs().max()).item()
        t = time_kernel(call) * 1e3
        out.append("%s:%x %.2f us (err %.1e)" % (name, tile if tile else lib.upp_linear_tile(M, N, K), t, err))
    print(" ".join(argv) or "base", "|", "  ".join(out))


if __name__ == "__main__":
    main()
