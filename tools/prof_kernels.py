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

from utils import synthetic as _seeded  # noqa: E402
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
# fused propagation step, adapter, frozen-branch row operators, Chamfer / EMD at the headline shapes
Lp, T, G2, D = 75, 64, 32, 384
X = torch.randn(B, Lp, D, device='cuda', requires_grad=True)
base = (torch.arange(B, device='cuda') * Lp + (Lp - T)).view(B, 1)
i1 = (base + torch.randint(0, T, (B, G2 * 8), device='cuda')).reshape(-1).int().contiguous()
i2 = (base + torch.stack([torch.randperm(T, device='cuda')[:G2] for _ in range(B)])).reshape(-1).int().contiguous()
idx8 = torch.randint(0, G2, (B, T, 8), device='cuda').int().contiguous()
w8 = torch.softmax(torch.randn(B, T, 8, device='cuda'), -1).contiguous()
bn = torch.nn.BatchNorm1d(D).cuda()
lnA = torch.nn.LayerNorm(D).cuda()
ha = torch.randn(B * Lp, D, device='cuda', requires_grad=True)
xa_ = torch.randn(B * Lp, D, device='cuda', requires_grad=True)
W1 = (torch.randn(32, D, device='cuda') * 0.05).requires_grad_(True); bb1 = torch.zeros(32, device='cuda', requires_grad=True)
W2 = (torch.randn(D, 32, device='cuda') * 0.05).requires_grad_(True); bb2 = torch.zeros(D, device='cuda', requires_grad=True)
ud = torch.rand(B * Lp, 32, device='cuda')
rows = torch.randn(B * 1096, 32, device='cuda')
bn32 = torch.nn.BatchNorm1d(32).cuda()
xyz1, xyz2 = torch.rand(B, 1096, 3, device='cuda'), torch.rand(B, 64, 3, device='cuda')
feat = torch.randn(B, 64, 32, device='cuda')
ca, cb = _seeded.unit_ball_clouds(B, 1024, seed=5).cuda().requires_grad_(True), _seeded.unit_ball_clouds(B, 1024, seed=6).cuda()
from models import upp_layers  # noqa: E402
# the commuted first propagation layer of the part-segmentation head: 1536-wide rows to 65,536 label points + rank-3 term
lp, cen128 = torch.rand(B, 2048, 3, device='cuda'), torch.rand(B, 128, 3, device='cuda')
z1536 = torch.randn(B, 128, 1536, device='cuda', requires_grad=True)
wt3 = torch.randn(3, 1536, device='cuda', requires_grad=True)
dseg, iseg = HF.sqdist_topk(lp, cen128, 3)
for _ in range(5):
    HF.interp_affine_train(dseg, iseg, z1536, lp, wt3, 3, 1e-4).sum().backward()
    index = HF.PropIndex(i1, i2, idx8, w8, B * Lp)
    HF.propagate(X, bn, index, None, 1.0, True).sum().backward()
    HF.adapter(ha, xa_, W1, bb1, W2, bb2, ud, 0.1, 0.7).sum().backward()
    HF.ln_adapter(X, X, None, None, 1.0, HF.ROW_STRIP_CLS, 10, lnA, W1, bb1, W2, bb2, None, 0.0, 0.7).sum().backward()   # block tail, one launch
    with torch.no_grad():
        HF.bn_rows(rows, bn32, True, relu=True)
        d_, i_ = upp_layers.square_distance(xyz1, xyz2).sort(dim=-1)
        HF.interp(d_, i_, feat, 16, 1e-4)
        HF.posenc(xyz1, [1.0, 2.0, 4.0, 8.0])
    d1, d2 = HF.ChamferFunction.apply(ca, cb)
    (d1.mean() + d2.mean()).backward()
    HF.EarthMoverDistanceFunction.apply(ca.detach(), cb).sum()
    idx, cen = ops.fps(x, 64, want_centers=True)
    _, _, nb = ops.knn(x, cen, 32, want_dist=False, want_neigh=True)
    ops.fps(x1228, 1024, want_centers=True)
    with torch.no_grad():
        enc(nb)
    xa, h = HF.rowln(tok, add=pos, prompts=prm, mode=HF.ROW_INSERT_CLS, P=10, gamma=g1, beta=b1)
    out = HF.attention(qkv, 6, 0.125)
    out.sum().backward()
# The Linear layers of the Transformer blocks at every token count of the headline step (bench.STEP_LINEAR_SHAPES) on the kernel the step uses
# for a frozen weight (csrc/linear_sb.hip), then the M = 2400 ones (bench.LINEAR_SHAPES) on the exact-f32 kernel (csrc/linear.hip).  Every labelled
# launch is preceded by a MARKER launch (transpose_kernel: used nowhere else in this script): tools/pmc_summary.py gives the first Linear
# dispatch behind the i-th marker the label LINEAR_ORDER[i % len] -- whatever else of the same kernel name ran before (round 3 counted
# dispatches modulo 8 and was rotated by five un-labelled launches of the patch embedding).
sys.path.insert(0, ROOT)
from bench import LINEAR_SHAPES, STEP_LINEAR_SHAPES, make_linear_operands, run_linear  # noqa: E402
gl = torch.Generator(device='cuda').manual_seed(11)
lin = [(make_linear_operands(ops, M, N, K, epi, torch.device('cuda'), gl), M, N, K, epi, ops.linear_sb_tile(M, N, K)) for _, M, N, K, epi in STEP_LINEAR_SHAPES]
marker = torch.zeros(8, 8, device='cuda')
for d_, M, N, K, epi, sb in lin:           # (plane images made before the labelled section: the split kernel of upp_linear_sb_prep is not a marker)
    run_linear(ops, d_, M, N, K, epi, sb)
plain = [(d_, M, N, K, epi, sb) for (d_, M, N, K, epi, sb), (lab, *_r) in zip(lin, STEP_LINEAR_SHAPES) if (lab, M, N, K, epi) in LINEAR_SHAPES]
for _ in range(6):
    for frozen, part in ((True, lin), (False, plain)):        # == pmc_summary.LINEAR_ORDER
        for d_, M, N, K, epi, sb in part:
            ops.transpose(marker)
            run_linear(ops, d_, M, N, K, epi, sb, frozen=frozen)
# round 3: the tall-matrix kernel and the grouped weight gradient at the segmentation head's largest layer (65,536 x 1536 -> 1024), the
# grouped weight gradients of one pre-training block, the denoising prompter's tail
xt = torch.randn(65536, 1536, device='cuda', generator=gl)
wt_ = torch.randn(1024, 1536, device='cuda', generator=gl) * 1536 ** -0.5
gt_ = torch.randn(65536, 1024, device='cuda', generator=gl)
pre = []
for M_, N_, K_ in ((2080, 1152, 384), (2080, 384, 384), (2080, 1536, 384), (2080, 384, 1536), (864, 1152, 384), (864, 384, 384), (864, 1536, 384), (864, 384, 1536)):
    pre.append((torch.randn(M_, N_, device='cuda', generator=gl), torch.randn(M_, K_, device='cuda', generator=gl)))
rf = torch.randn(B, 1096, 32, device='cuda', generator=gl)
rw0, rb0 = torch.randn(64, 32, device='cuda', generator=gl) * 0.2, torch.zeros(64, device='cuda')
rw1, rb1 = torch.randn(3, 64, device='cuda', generator=gl) * 0.2, torch.zeros(3, device='cuda')
for _ in range(4):
    ops.linear_f32(xt, wt_)
    ops.linear_wgrad_grouped([(gt_, xt)])
    ops.linear_wgrad_grouped(pre)
    ops.rectify_select(rf, rw0, rb0, rw1, rb1, xyz1, 972)
torch.cuda.synchronize()
print("done")
