"""Where a upp_linear_f32 launch spends its time: a diagnostic build of csrc/linear.hip (-DUPP_LIN_STAMPS) whose wave 0 of
every workgroup stamps s_memtime at kernel entry / first MFMA / end of the k loop / after the stores, plus the constant
100 MHz s_memrealtime at both ends (-> in-kernel shader clock).  Not part of the product library."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402
from upp_hip.build import FLAGS as PRODUCT_FLAGS  # noqa: E402  (the flag list of the shipped library, -packed-fp32-ops included: a micro build measures the same code)

SRC = os.path.join(ROOT, "iccv2025-upp_amd", "upp_hip", "csrc", "linear.hip")
SO = "/tmp/liblin_stamps.so"


def main():
    extra = ["-D" + a for a in sys.argv[1:]]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + PRODUCT_FLAGS + ["-shared",
                           "-DUPP_LIN_STAMPS"] + extra + [SRC, os.path.join(os.path.dirname(SRC), "linear_rt.hip"), os.path.join(os.path.dirname(SRC), "abi.hip"), "-o", SO])
    lib = ctypes.CDLL(SO)
    vp, ll, ci = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
    lib.upp_linear_f32.argtypes = [vp, ll, vp, ll, vp, vp, ll, vp, ll, ci, ci, ci, ci, ci, vp]
    lib.upp_linear_set_stamps.argtypes = [vp]
    dev = torch.device("cuda", 0)
    stamps = torch.zeros(1024 * 8, dtype=torch.int64, device=dev)
    lib.upp_linear_set_stamps(stamps.data_ptr())
    shapes = [("fc1_2400", 2400, 1536, 384, 0x4412), ("qkv_2400", 2400, 1152, 384, 0x4311), ("proj_2400", 2400, 384, 384, 0x2241),
              ("fc2_2400", 2400, 384, 1536, 0x2241), ("fc1_1120", 1120, 1536, 384, 0x2421)]
    for name, M, N, K, tile in shapes:
        a = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        c = torch.empty(M, N, device=dev)
        for _ in range(int(os.environ.get('LAUNCHES', '30'))):        # back-to-back launches: the last one is read
            rc = lib.upp_linear_f32(a.data_ptr(), K, w.data_ptr(), K, None, c.data_ptr(), N, None, 0, M, N, K, 0, tile, None)
            assert rc == 0, rc
        torch.cuda.synchronize()
        bmb, bnb = tile >> 12, (tile >> 8) & 15
        nwg = -(-M // (32 * bmb)) * -(-N // (32 * bnb))
        s = stamps[:nwg * 8].view(nwg, 8).cpu().double()
        t0, t1, t2, t3, r0, r1 = (s[:, i] for i in range(6))
        clk = ((t3 - t0) / ((r1 - r0) * 10.0)).median().item()                  # cycles per ns = GHz
        us = lambda d: (d / clk / 1e3)                                           # noqa: E731
        span = (s[:, 5].max() - s[:, 4].min()).item() * 0.01
        print("%-10s tile %x wgs %3d | clock %.2f GHz | prologue %.2f us, k-loop %.2f us (%.0f cycles), epilogue %.2f us | whole grid %.2f us"
              % (name, tile, nwg, clk, us(t1 - t0).median(), us(t2 - t1).median(), (t2 - t1).median(), us(t3 - t2).median(), span), flush=True)


if __name__ == "__main__":
    main()
