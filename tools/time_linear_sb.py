"""upp_linear_sb_f32 (split-bf16, csrc/linear_sb.hip) against upp_linear_f32 (exact-f32 MFMA) at the Transformer-block shapes: error against
float64 and device time per call (HIP-graph replay, HIP events).

    python tools/time_linear_sb.py [--tiles] [--rows M[,M...]] [--out gpurun_out/time_linear_sb.jsonl]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]

import torch  # noqa: E402

from bench import time_kernel  # noqa: E402
from upp_hip import ops, _abi  # noqa: E402

LAYERS = (("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536), ("dqkv", 384, 1152))
SB_TILES = [0x400000 + a * 65536 + b * 4096 + c * 256 + d * 16 + e for (a, b, c, d, e) in
            [(8, 4, 4, 1, 2), (8, 4, 2, 1, 2), (4, 4, 2, 1, int(os.environ.get("UPP_SB_NST44", "3"))), (4, 3, 1, 1, 4), (3, 4, 2, 1, 4), (2, 4, 2, 1, 4), (2, 3, 1, 1, 4), (2, 2, 1, 2, 3), (2, 2, 2, 4, 2), (1, 2, 1, 2, 4)]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", action="store_true")
    ap.add_argument("--rows", default="2400,2080,2048,1120")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "time_linear_sb.jsonl"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = _abi.load()
    g = torch.Generator(device=dev).manual_seed(0)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    f = open(args.out, "w")
    for M in [int(m) for m in args.rows.split(",")]:
        for name, N, K in LAYERS:
            a = torch.randn(M, K, device=dev, generator=g)
            w = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
            w._upp_persistent = True            # (a plain tensor as a frozen weight: let ops.PLANES keep its plane image)
            b = torch.randn(N, device=dev, generator=g)
            ref = a.double() @ w.double().t()
            bound = a.abs().double() @ w.abs().double().t()
            out = torch.empty(M, N, device=dev)
            o32 = ops.linear_f32(a, w)
            osb = ops.linear_f32(a, w, frozen=True)
            planes = ops.PLANES.get(w)
            flop = 2.0 * M * N * K
            t32 = time_kernel(lambda: ops.linear_f32(a, w, out=out))
            tsb = time_kernel(lambda: ops.linear_f32(a, w, out=out, frozen=True))
            t32g = time_kernel(lambda: ops.linear_f32(a, w, b, ops.LIN_BIAS_GELU_D))
            tsbg = time_kernel(lambda: ops.linear_f32(a, w, b, ops.LIN_BIAS_GELU_D, frozen=True))
            row = {"shape": "%s_%d" % (name, M), "M": M, "N": N, "K": K, "tile_f32": "%x" % lib.upp_linear_tile(M, N, K),
                   "tile_sb": "%x" % ops.linear_sb_tile(M, N, K), "us_f32": t32 * 1e3, "us_sb": tsb * 1e3, "us_f32_gelu_d": t32g * 1e3,
                   "us_sb_gelu_d": tsbg * 1e3, "tf_f32": flop / t32 / 1e9, "tf_sb": flop / tsb / 1e9,
                   "err_f32": ((o32.double() - ref).abs() / bound).max().item(), "err_sb": ((osb.double() - ref).abs() / bound).max().item()}
            if args.tiles:
                for t in SB_TILES:
                    if K % (32 * ((t >> 4) & 15)) or K // (32 * ((t >> 4) & 15)) < (t & 15):
                        continue
                    bm, bn = (t >> 16) & 15, (t >> 12) & 15
                    if ((M + 32 * bm - 1) // (32 * bm)) * ((N + 32 * bn - 1) // (32 * bn)) > 1024:
                        continue

                    def run(t=t):
                        ops._call(dev, "upp_linear_sb_f32", _abi.ptr(a), K, _abi.ptr(planes), None, _abi.ptr(out), N, None, 0, M, N, K, 0, t)
                    row["us_tile_%x" % t] = time_kernel(run) * 1e3
            print(json.dumps(row), flush=True)
            f.write(json.dumps(row) + "\n")
    f.close()


if __name__ == "__main__":
    main()
