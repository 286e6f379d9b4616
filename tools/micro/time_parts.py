"""upp_linear_f32 (contraction inside one workgroup) against upp_linear_parts_f32 (contraction cut over workgroups, partial outputs) at the
narrow-output shapes of the Transformer blocks: us per launch (HIP-graph replay)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "iccv2025-upp_amd")]
import torch
from bench import time_kernel
from upp_hip import ops, _abi
lib = _abi.load()
for M in (2400, 2080, 2048, 1120, 4128, 4448):
    row = []
    for name, N, K in (("fc2", 384, 1536), ("dqkv", 384, 1152), ("proj", 384, 384)):
        a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * K ** -0.5
        out = torch.empty(M, N, device='cuda')
        t0 = time_kernel(lambda: ops.linear_f32(a, w, out=out)) * 1e3
        res = ["%s %5.1f" % (name, t0)]
        for P in sorted({lib.upp_linear_parts(M, N, K), 2, 3, 4, 6, 8}):
            if P < 2 or K % (32 * P):
                continue
            try:
                t = time_kernel(lambda: ops.linear_parts(a, w, P)) * 1e3
            except RuntimeError:
                continue
            res.append("P%d%s %5.1f" % (P, "*" if P == lib.upp_linear_parts(M, N, K) else "", t))
        row.append(" ".join(res))
    print("M=%4d | " % M + " | ".join(row), flush=True)
