"""upp_linear_f32 against the library GEMM (torch F.linear -> hipBLASLt) at the Transformer-block shapes:
correctness vs an f64 reference and device time per call (HIP-graph replay, HIP events).

    python tools/time_linear.py [--tiles] [--tuned] [--rows M[,M...]]     (--rows: the five block shapes at other token counts)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from bench import time_kernel  # noqa: E402
from upp_hip import ops, _abi  # noqa: E402

SHAPES = [  # (name, M, N, K)
    ("qkv_2400", 2400, 1152, 384), ("proj_2400", 2400, 384, 384), ("fc1_2400", 2400, 1536, 384), ("fc2_2400", 2400, 384, 1536),
    ("dqkv_2400", 2400, 384, 1152),
    ("qkv_2080", 2080, 1152, 384), ("proj_2080", 2080, 384, 384), ("fc1_2080", 2080, 1536, 384), ("fc2_2080", 2080, 384, 1536),
    ("qkv_2048", 2048, 1152, 384), ("fc1_2048", 2048, 1536, 384), ("fc2_2048", 2048, 384, 1536),
    ("qkv_1120", 1120, 1152, 384), ("proj_1120", 1120, 384, 384), ("fc1_1120", 1120, 1536, 384), ("fc2_1120", 1120, 384, 1536),
]
TILES = [0x4412, 0x4311, 0x3411, 0x2421, 0x2321, 0x2241, 0x1241, 0x2221, 0x2512211]      # the last: linear_rt.hip's register-tiled kernel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", action="store_true", help="time every wave-tile shape, not only the library's choice")
    ap.add_argument("--tuned", action="store_true", help="library GEMM with TunableOp-selected solutions")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "time_linear.json"))
    ap.add_argument("--rows", default="", help="comma-separated row counts: time qkv / proj / fc1 / fc2 / dqkv at these instead")
    args = ap.parse_args()
    global SHAPES
    if args.rows:
        SHAPES = [("%s_%d" % (n, int(m)), int(m), N, K) for m in args.rows.split(",")
                  for n, N, K in (("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536), ("dqkv", 384, 1152))]
    dev = torch.device("cuda", 0)
    if args.tuned:                      # the library side of the comparison with TunableOp-selected solutions (measurement only)
        os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"; os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
        torch.cuda.tunable.enable(True)
    lib = _abi.load()
    g = torch.Generator(device=dev).manual_seed(0)
    rows = []
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) * 0.05
        b = torch.randn(N, device=dev, generator=g)
        ref = (a.double() @ w.double().t())
        out = ops.linear_f32(a, w)
        lib_out = F.linear(a, w)
        scale = ref.abs().max().item()
        err = (out.double() - ref).abs().max().item() / scale
        err_lib = (lib_out.double() - ref).abs().max().item() / scale
        zg, dg = ops.linear_f32(a, w, b, ops.LIN_BIAS_GELU_D)
        zr = ref + b.double()
        gref = 0.5 * zr * (1 + torch.erf(zr / 2 ** 0.5))
        err_g = (zg.double() - gref).abs().max().item() / gref.abs().max().item()
        t_ours = time_kernel(lambda: ops.linear_f32(a, w, out=out))
        t_lib = time_kernel(lambda: F.linear(a, w))
        t_gelu = time_kernel(lambda: ops.linear_f32(a, w, b, ops.LIN_BIAS_GELU_D))
        flop = 2.0 * M * N * K
        tile = lib.upp_linear_tile(M, N, K)
        row = {"shape": name, "M": M, "N": N, "K": K, "tile": "%x" % tile, "us_ours": t_ours * 1e3,
               "us_lib": t_lib * 1e3, "us_ours_gelu_d": t_gelu * 1e3, "tf_ours": flop / t_ours / 1e9, "tf_lib": flop / t_lib / 1e9,
               "err_ours": err, "err_lib": err_lib, "err_gelu": err_g}
        if args.tiles:
            for t in TILES:
                if not (t & 0x10000) and K % (32 * ((t >> 4) & 15) * (t & 15)):
                    continue
                row["us_tile_%x" % t] = time_kernel(lambda: ops.linear_f32(a, w, out=out, tile=t)) * 1e3
        rows.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
