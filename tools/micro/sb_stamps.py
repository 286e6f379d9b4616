"""Where a upp_linear_sb_f32 launch spends its time: a diagnostic build of csrc/linear_sb.hip (-DUPP_LIN_STAMPS [-D...]) whose thread 0 of
every workgroup stamps s_memtime at kernel entry / loop entry / loop exit / after the stores, plus s_memrealtime at both ends.
    python tools/micro/sb_stamps.py [UPP_SB_NO_DMA ...]
Not part of the product library."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch  # noqa: E402
from upp_hip.build import FLAGS as PRODUCT_FLAGS  # noqa: E402  (the flag list of the shipped library, -packed-fp32-ops included: a micro build measures the same code)

CSRC = os.path.join(ROOT, "iccv2025-upp_amd", "upp_hip", "csrc")
SO = "/tmp/libsb_stamps.so"


def main():
    extra = ["-D" + a for a in sys.argv[1:]]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + PRODUCT_FLAGS + ["-shared",
                           "-DUPP_LIN_STAMPS"] + extra + [os.path.join(CSRC, "linear_sb.hip"), os.path.join(CSRC, "abi.hip"), "-o", SO])
    lib = ctypes.CDLL(SO)
    vp, ll, ci = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
    lib.upp_linear_sb_f32.argtypes = [vp, ll, vp, vp, vp, ll, vp, ll, ci, ci, ci, ci, ci, vp]
    lib.upp_linear_sb_prep.argtypes = [vp, ll, ci, ci, ci, vp, vp]
    lib.upp_linear_sb_planes_bytes.restype = ll
    lib.upp_linear_sb_set_stamps.argtypes = [vp]
    dev = torch.device("cuda", 0)
    stamps = torch.zeros(1024 * 8, dtype=torch.int64, device=dev)
    lib.upp_linear_sb_set_stamps(stamps.data_ptr())
    shapes = [("fc1_2400", 2400, 1536, 384), ("qkv_2400", 2400, 1152, 384), ("proj_2400", 2400, 384, 384), ("fc2_2400", 2400, 384, 1536),
              ("fc1_1120", 1120, 1536, 384), ("fc2_1120", 1120, 384, 1536)]
    print("variant", extra)
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        c = torch.empty(M, N, device=dev)
        planes = torch.empty(int(lib.upp_linear_sb_planes_bytes(N, K)), dtype=torch.uint8, device=dev)
        assert lib.upp_linear_sb_prep(w.data_ptr(), K, N, K, 0, planes.data_ptr(), None) == 0
        tile = lib.upp_linear_sb_tile(M, N, K)
        for _ in range(int(os.environ.get('LAUNCHES', '2000'))):        # back-to-back launches (clock ramp): the last one is read
            rc = lib.upp_linear_sb_f32(a.data_ptr(), K, planes.data_ptr(), None, c.data_ptr(), N, None, 0, M, N, K, 0, tile, None)
            assert rc == 0, rc
        torch.cuda.synchronize()
        bmb, bnb = (tile >> 16) & 15, (tile >> 12) & 15
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
