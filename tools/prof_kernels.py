"""Launch each main hand-written kernel a few times at the headline shapes (B=32), stand-alone, so that
rocprofv3 --kernel-trace / --pmc rows can be attributed per kernel.
    rocprofv3 --kernel-trace --stats ... -- python tools/prof_kernels.py
    rocprofv3 --pmc FETCH_SIZE ... -- python tools/prof_kernels.py      (and a second pass with WRITE_SIZE)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "tests", "iccv2025-upp_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402

import _seeded  # noqa: E402
from models.upp_layers import Encoder  # noqa: E402
from upp_hip import functional as HF, ops  # noqa: E402

B = 32
torch.manual_seed(0)
x = _seeded.unit_ball_clouds(B, 1024, seed=1).cuda()
x1228 = _seeded.unit_ball_clouds(B, 1228, seed=3).cuda()
enc = Encoder(384).cuda().train()
for p in enc.parameters():
    p.requires_grad_(False)
tok = torch.randn(B, 65, 384, device='cuda')
pos = torch.randn(B, 65, 384, device='cuda')
prm = torch.randn(10, 384, device='cuda')
g1, b1 = torch.ones(384, device='cuda'), torch.zeros(384, device='cuda')
qkv = torch.randn(B, 75, 1152, device='cuda', requires_grad=True)
for _ in range(5):
    idx, cen = ops.fps(x, 64, want_centers=True)
    _, _, nb = ops.knn(x, cen, 32, want_dist=False, want_neigh=True)
    ops.fps(x1228, 1024, want_centers=True)
    with torch.no_grad():
        enc(nb)
    xa, h = HF.rowln(tok, add=pos, prompts=prm, mode=HF.ROW_INSERT_CLS, P=10, gamma=g1, beta=b1)
    out = HF.attention(qkv, 6, 0.125)
    out.sum().backward()
torch.cuda.synchronize()
print("done")
