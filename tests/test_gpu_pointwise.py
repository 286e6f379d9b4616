"""Row operators of the frozen prompter branches (csrc/pointwise.hip) against the torch ops they replace."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from models import upp_layers
from upp_hip import functional as HF

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-5, atol_scale=2e-6):
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_scale * max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("R,C", [(35072, 32), (35072, 64), (4099, 64), (65536, 128), (4101, 128), (16384, 12), (1024, 64), (32, 256), (7, 3), (2400, 384), (1000, 300),
                                 (5, 512)])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("relu", [True, False])
def test_bn_rows_matches_torch_batch_norm(R, C, training, relu):
    torch.manual_seed(R + C)
    x = (torch.randn(R, C, device='cuda') * torch.linspace(0.2, 3.0, C, device='cuda') + torch.linspace(-8.0, 5.0, C, device='cuda'))
    outs = []
    for fused in (True, False):
        bn = torch.nn.BatchNorm1d(C).cuda()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
            bn.running_mean.copy_(torch.linspace(-1, 1, C)); bn.running_var.copy_(torch.linspace(0.5, 2, C))
            if fused:
                y = HF.bn_rows(x, bn, training, relu)
            else:
                y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)
                y = F.relu(y) if relu else y
        outs.append((y, bn.running_mean.clone(), bn.running_var.clone()))
    # column offsets of up to 40 sigma: an unshifted sum-of-squares variance would lose most digits here
    close(outs[0][0], outs[1][0], rtol=2e-5, atol_scale=5e-6)
    close(outs[0][1], outs[1][1], rtol=1e-5, atol_scale=1e-6)
    close(outs[0][2], outs[1][2], rtol=2e-5, atol_scale=1e-6)
    with torch.no_grad():
        bn2 = torch.nn.BatchNorm1d(C).cuda()
        assert torch.equal(HF.bn_rows(x, bn2, training, relu), HF.bn_rows(x, torch.nn.BatchNorm1d(C).cuda(), training, relu))   # deterministic


@pytest.mark.parametrize("R,C", [(65536, 512), (16384, 1536), (4096, 128), (4101, 128), (4099, 64), (1000, 300), (32, 64), (7, 3), (2400, 384)])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("x_grad", [True, False])
def test_bn_rows_train_backward_matches_torch(R, C, relu, x_grad):
    """upp_bn_rows_fwd + upp_bn_rows_bwd (trainable per-point heads) against torch's batch_norm autograd."""
    torch.manual_seed(R + C)
    x0 = (torch.randn(R, C, device='cuda') * torch.linspace(0.2, 3.0, C, device='cuda') + torch.linspace(-4.0, 3.0, C, device='cuda'))
    gy = torch.randn(R, C, device='cuda') * torch.linspace(0.5, 2.0, C, device='cuda')
    res = []
    mask = None
    for fused in (True, False):
        bn = torch.nn.BatchNorm1d(C).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
        x = x0.clone().requires_grad_(x_grad)
        if fused:
            y = upp_layers._bn_rows(x, bn, True, relu=relu)
            assert type(y.grad_fn).__name__ == '_BnRowsTrainBackward'
            y.backward(gy)
            mask = (y > 0).float() if relu else None
        else:
            # the ReLU gate of the handful of outputs within rounding of 0 depends on the evaluation order of the affine
            # transform: take the gate from the fused forward, so the comparison is about the BatchNorm backward
            y_lin = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps)
            y = F.relu(y_lin) if relu else y_lin
            y_lin.backward(gy * mask if relu else gy)
        res.append((y, x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()))
    (y, gx, gg, gb, rm, rv), (y_t, gx_t, gg_t, gb_t, rm_t, rv_t) = res
    close(y, y_t, rtol=2e-5, atol_scale=5e-6)
    close(rm, rm_t, rtol=1e-5, atol_scale=1e-6); close(rv, rv_t, rtol=2e-5, atol_scale=1e-6)
    close(gb, gb_t, rtol=2e-5, atol_scale=2e-5)          # sums of R terms of either sign: absolute error ~ sqrt(R) * eps * |g|
    close(gg, gg_t, rtol=2e-5, atol_scale=2e-5)
    if x_grad:
        close(gx, gx_t, rtol=5e-5, atol_scale=1e-5)
    else:
        assert gx is None


@pytest.mark.parametrize("B,N,S,C,k", [(32, 1096, 64, 32, 16), (32, 64, 32, 12, 16), (2, 50, 5, 7, 16), (3, 33, 40, 256, 3), (1, 1, 16, 1, 1)])
def test_interp_matches_reference_formula(B, N, S, C, k):
    g = torch.Generator(device='cuda').manual_seed(B * N + C)
    xyz1 = torch.rand(B, N, 3, device='cuda', generator=g)
    xyz2 = torch.rand(B, S, 3, device='cuda', generator=g)
    if S >= 4 and N >= 4:
        xyz2[:, :4] = xyz1[:, :4]                                         # coincident points: d ~ 0 (+- rounding)
    feat = torch.randn(B, S, C, device='cuda', generator=g)
    dists, idx = upp_layers.square_distance(xyz1, xyz2).sort(dim=-1)
    kk = min(k, S)
    d, i = dists[:, :, :kk], idx[:, :, :kk]
    recip = 1.0 / (d + 1e-4)
    w = recip / recip.sum(dim=2, keepdim=True)
    want = torch.sum(upp_layers.index_points(feat, i) * w.unsqueeze(-1), dim=2)
    got = HF.interp(dists, idx, feat, kk, 1e-4)
    close(got, want, rtol=1e-5, atol_scale=2e-6)
    wide = torch.full((B, N, C + 5), 7.0, device='cuda')
    HF.interp(dists, idx, feat, kk, 1e-4, out=wide, col0=5)
    assert torch.equal(wide[:, :, 5:], got) and bool((wide[:, :, :5] == 7.0).all())
    # and through the layer helper (HIP tensors without grad: upp_sqdist_topk + upp_interp_fwd).  Its distances differ from the
    # matmul's in the last bit; a weight 1/(d + 1e-4) at a coincident point (d ~ 1e-7 of rounding noise) amplifies that 1000x
    # -- in the reference as well -- so the first four rows get a correspondingly wider tolerance.
    via = upp_layers._inverse_distance_interp(xyz1, xyz2, feat, k, 1e-4)
    lo = 4 if (S >= 4 and N >= 4) else 0
    close(via[:, lo:], want[:, lo:], rtol=2e-4, atol_scale=2e-4)
    if lo:
        close(via[:, :lo], want[:, :lo], rtol=2e-2, atol_scale=2e-2)


@pytest.mark.parametrize("rows,F_", [((32, 1096), 4), ((5,), 8), ((2, 3, 7), 1), ((4, 4), 0)])
def test_posenc_matches_sin_cos_concat(rows, F_):
    x = (torch.rand(*rows, 3, device='cuda') - 0.5) * 2.5
    pe = upp_layers.PositionalEmbedding(max(F_, 1))
    freqs = pe.freq_bands[:F_]
    want = torch.cat([x] + [f(freq * x) for freq in freqs for f in (torch.sin, torch.cos)], -1)
    got = HF.posenc(x, freqs)
    assert got.shape == want.shape
    close(got, want, rtol=1e-6, atol_scale=1e-7)
    wide = torch.zeros(*rows, want.shape[-1] + 9, device='cuda')
    HF.posenc(x, freqs, out=wide, col0=0)
    assert torch.equal(wide[..., :want.shape[-1]], got) and bool((wide[..., want.shape[-1]:] == 0).all())
    if F_ >= 1:
        close(pe(x), torch.cat([x] + [f(freq * x) for freq in pe.freq_bands for f in (torch.sin, torch.cos)], -1), rtol=1e-6, atol_scale=1e-7)


def test_rectify_prompter_fused_path_equals_torch_path():
    from models import build_model_from_cfg
    from utils.config import builtin_cfg
    import _seeded
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda()
    for p in m.parameters():
        p.requires_grad_(False)
    rp = m.rectify_prompter
    pts = _seeded.noisy_clouds(4, 1024, seed=5).cuda()
    with torch.no_grad():
        _, center = m.group_divider(pts)
    c32 = center[:, :32].contiguous()
    tok = torch.randn(4, 32, 384, device='cuda')
    for training in (False, True):
        rp.train(training)
        state = {k: v.clone() for k, v in rp.state_dict().items()}
        torch.manual_seed(0)
        fused = rp(pts, c32, tok)                                  # nothing requires grad -> fused operators
        stats_fused = {k: v.clone() for k, v in rp.state_dict().items()}
        rp.load_state_dict(state)
        torch.manual_seed(0)
        tok_g = tok.clone().requires_grad_(True)                   # an input that requires grad forces the torch ops
        plain = rp(pts, c32, tok_g)
        if not training:                                           # (train mode draws a dropout mask: statistics only)
            close(fused, plain, rtol=1e-4, atol_scale=2e-5)
        for k in state:
            if 'running' in k:
                close(stats_fused[k], rp.state_dict()[k], rtol=1e-4, atol_scale=1e-5)
            elif 'num_batches' in k:
                assert int(stats_fused[k]) == int(rp.state_dict()[k])


@pytest.mark.parametrize("pending", [False, True])
@pytest.mark.parametrize("R,C,p", [(65536, 512, 0.5), (8192, 256, 0.2), (8190, 128, 0.3), (1000, 300, 0.5), (4096, 40, 0.7)])
def test_bn_relu_dropout_in_the_batchnorm_passes(R, C, p, pending):
    """Round 6: `BatchNorm1d, ReLU, Dropout(p)` of the segmentation head (reference models/Point_MAE_unify_segment.py:424-427) in the
    BatchNorm's own passes (upp_bn_rows_drop_fwd / _bwd): kept values are relu(bn(x)) / (1 - p) exactly, the keep rate is 1 - p, the mask
    is a function of (num_batches_tracked, element) -- the same for forward and backward of a step, another one after the counter moved,
    nothing stored -- and the gradients equal torch's for that mask.  Tall kernels (C % 256 == 0) and flat ones.  pending: inside a model
    forward the counter bump is queued for the end of the forward (upp_layers.end_forward), i.e. it lands BETWEEN this layer's forward and
    its backward: both must still see one mask."""
    torch.manual_seed(R + C)
    x0 = torch.randn(R, C, device='cuda') * torch.linspace(0.5, 2.0, C, device='cuda') + torch.linspace(-1.0, 1.0, C, device='cuda')
    gy = torch.randn(R, C, device='cuda')
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    drop = torch.nn.Dropout(p).train()
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
    if pending:
        upp_layers.begin_forward(x0.device, True)
    upp_layers.bump_counter(bn.num_batches_tracked)
    x = x0.clone().requires_grad_(True)
    y = upp_layers._bn_rows(x, bn, True, relu=True, drop=drop)
    assert type(y.grad_fn).__name__ == '_BnRowsTrainBackward'
    if pending:
        assert int(bn.num_batches_tracked) == 0
        upp_layers.end_forward()
    assert int(bn.num_batches_tracked) == 1
    y.backward(gy)
    bn_ref = torch.nn.BatchNorm1d(C).cuda().train()
    with torch.no_grad():
        bn_ref.weight.copy_(bn.weight); bn_ref.bias.copy_(bn.bias)
        plain = upp_layers._bn_rows(x0, bn_ref, True, relu=True)          # relu(bn(x)) on the same kernels, no dropout
    pos = plain > 0
    kept = (y != 0)
    assert not (kept & ~pos).any()
    rate = kept[pos].float().mean().item()
    assert abs(rate - (1.0 - p)) < 4.0 * (p * (1 - p) / pos.sum().item()) ** 0.5 + 1e-3, rate
    assert torch.equal(y[kept], (plain * (1.0 / (1.0 - p)))[kept])
    col_rate = (kept & pos).float().sum(0) / pos.float().sum(0).clamp_min(1.0)          # no column (or row) is systematically kept / dropped
    assert (col_rate - (1.0 - p)).abs().max().item() < 0.2 if R >= 4096 else True
    # the same counter value -> the same mask; another value -> another mask
    with torch.no_grad():
        rm, rv = bn.running_mean.clone(), bn.running_var.clone()
        y2 = upp_layers._bn_rows(x0, bn, True, relu=True)                   # (no gradient asked: the frozen-branch path, no dropout here)
        bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    x2 = x0.clone().requires_grad_(True)
    y_same = upp_layers._bn_rows(x2, bn, True, relu=True, drop=drop)
    assert torch.equal(y_same != 0, kept)
    upp_layers.bump_counter(bn.num_batches_tracked)
    y_next = upp_layers._bn_rows(x0.clone().requires_grad_(True), bn, True, relu=True, drop=drop)
    differ = ((y_next != 0) != kept)[pos].float().mean().item()
    assert abs(differ - 2 * p * (1 - p)) < 0.02, differ
    # gradients against torch for that mask
    mask = torch.where(pos, kept.float(), torch.ones_like(plain)) / (1.0 - p)
    xt = x0.clone().requires_grad_(True)
    y_lin = F.batch_norm(xt, None, None, bn_ref.weight, bn_ref.bias, True, 0.1, bn.eps)
    (y_lin * pos.float() * mask).backward(gy)
    close(x.grad, xt.grad, rtol=5e-5, atol_scale=1e-5)
    close(bn.weight.grad, bn_ref.weight.grad, rtol=2e-5, atol_scale=2e-5)
    close(bn.bias.grad, bn_ref.bias.grad, rtol=2e-5, atol_scale=2e-5)
    # eval mode / p = 0: the module's identity
    bn.eval(); drop.eval()
    with torch.no_grad():
        assert torch.equal(upp_layers._bn_rows(x0, bn, False, relu=True, drop=drop), upp_layers._bn_rows(x0, bn, False, relu=True))


@pytest.mark.parametrize("B,N,S,C,k", [(8, 2048, 128, 1152, 3), (4, 1076, 32, 32, 16), (2, 32, 32, 12, 16), (3, 33, 40, 300, 3), (1, 1, 16, 1, 1),
                                       (2, 4096, 5, 7, 5), (2, 64, 200, 96, 6), (2, 300, 150, 70, 4), (2, 500, 64, 64, 8), (3, 257, 16, 20, 5), (2, 1000, 128, 32, 3)])
def test_interp_train_forward_and_feature_gradient(B, N, S, C, k):
    """upp_interp_fwd / upp_interp_bwd (trainable features, constant geometry) against the reference's torch formula."""
    g = torch.Generator(device='cuda').manual_seed(B * N + C)
    xyz1 = torch.rand(B, N, 3, device='cuda', generator=g)
    xyz2 = torch.rand(B, S, 3, device='cuda', generator=g)
    feat0 = torch.randn(B, S, C, device='cuda', generator=g)
    gy = torch.randn(B, N, C, device='cuda', generator=g)
    outs = []
    for fused in (True, False):
        feat = feat0.clone().requires_grad_(True)
        dists, idx = upp_layers.square_distance(xyz1, xyz2).sort(dim=-1)      # the same neighbour table on both sides
        if fused:
            y = HF.interp_train(dists, idx, feat, k, 1e-4)
            assert type(y.grad_fn).__name__ == '_InterpTrainBackward'
        else:
            d, i = dists[:, :, :k], idx[:, :, :k]
            recip = 1.0 / (d + 1e-4)
            w = recip / recip.sum(dim=2, keepdim=True)
            y = torch.sum(upp_layers.index_points(feat, i) * w.unsqueeze(-1), dim=2)
        y.backward(gy)
        outs.append((y, feat.grad))
    close(outs[0][0], outs[1][0], rtol=1e-5, atol_scale=2e-6)
    close(outs[0][1], outs[1][1], rtol=2e-5, atol_scale=5e-6)
    feat = feat0.clone().requires_grad_(True)
    HF.interp_train(dists, idx, feat, k, 1e-4).backward(gy)
    assert torch.equal(feat.grad, outs[0][1])                           # deterministic
    feat = feat0.clone().requires_grad_(True)                           # the layer helper routes trainable features here
    y = upp_layers._inverse_distance_interp(xyz1, xyz2, feat, k, 1e-4)
    assert type(y.grad_fn).__name__ == '_InterpTrainBackward'
    close(y, outs[1][0], rtol=1e-3, atol_scale=1e-3)


@pytest.mark.parametrize("B,N,S,k", [(32, 1096, 64, 16), (32, 2048, 128, 3), (4, 96, 32, 16), (3, 50, 200, 6), (2, 33, 256, 16), (2, 7, 5, 5),
                                     (1, 1, 1, 1), (2, 40, 65, 8)])
def test_sqdist_topk_matches_sorted_square_distance(B, N, S, k):
    """upp_sqdist_topk against square_distance().sort() (reference models/modules.py:13-32 + Point_MAE_unify.py:34-36)."""
    g = torch.Generator(device='cuda').manual_seed(B * N + S)
    xyz1 = torch.rand(B, N, 3, device='cuda', generator=g) * 2 - 1
    xyz2 = torch.rand(B, S, 3, device='cuda', generator=g) * 2 - 1
    if S >= 8:
        xyz2[:, 5] = xyz2[:, 2]                                      # exact duplicates: the lower index must come first
    d, i = HF.sqdist_topk(xyz1, xyz2, k)
    assert d.shape == (B, N, k) and i.dtype == torch.int64
    ref = upp_layers.square_distance(xyz1.double(), xyz2.double())
    rd, ri = ref.sort(dim=-1, stable=True)
    close(d, rd[:, :, :k].float(), rtol=1e-5, atol_scale=1e-6)
    assert (d[:, :, 1:] >= d[:, :, :-1]).all()
    # identical neighbour lists wherever the float64 distances are separated by more than f32 rounding
    gap = (rd[:, :, 1:k + 1] - rd[:, :, :k]) if S > k else torch.full_like(rd[:, :, :k], 1.0)
    if S > k:
        clear = (gap.abs().min(dim=-1)[0] > 1e-5) & ((rd[:, :, 1:k] - rd[:, :, :k - 1]).abs().min(dim=-1)[0] > 1e-5 if k > 1 else True)
    else:
        clear = (rd[:, :, 1:k] - rd[:, :, :k - 1]).abs().min(dim=-1)[0] > 1e-5 if k > 1 else torch.ones(B, N, dtype=torch.bool, device='cuda')
    if S >= 8:      # rows that select one of the duplicated points: order (2 before 5) is checked separately
        dup = ((i == 2) | (i == 5)).any(-1)
        both = (i == 2).any(-1) & (i == 5).any(-1)
        pos2 = (i == 2).float().argmax(-1); pos5 = (i == 5).float().argmax(-1)
        assert (pos2[both] + 1 == pos5[both]).all()
        clear = clear & ~dup
    assert clear.float().mean() > 0.1 or B * N < 10
    assert torch.equal(i[clear], ri[:, :, :k][clear])
    # and the table feeds the interpolation exactly like the sorted one
    feat = torch.randn(B, S, 20, device='cuda', generator=g)
    td, ti = upp_layers.square_distance(xyz1, xyz2).sort(dim=-1)
    a, bb = HF.interp(d, i, feat, k, 1e-4), HF.interp(td, ti, feat, k, 1e-4)
    if clear.any():      # (rows whose k-th neighbour is one of the duplicates may keep either copy: torch.sort is not stable)
        close(a[clear], bb[clear], rtol=2e-4, atol_scale=2e-4)


@pytest.mark.parametrize("B,N,S,Cin,mlp,k", [(2, 512, 64, 96, [128, 64], 3), (3, 2048, 128, 1152, [1536, 1024], 3)])
def test_feature_propagation_commuted_first_layer_equals_the_concat_formulation(B, N, S, Cin, mlp, k):
    """W.[p1 | interp(x)] + b == W_p.p1 + interp(W_x.x + b): outputs, running statistics and gradients (trainable layer, as
    in the part-segmentation head) of the commuted form are as close to an f64 evaluation of the reference formulation as
    the f32 concat formulation is (ReLU gates that flip under rounding make both differ from f64 by the same amount)."""
    torch.manual_seed(5)
    dev = 'cuda'
    xyz1 = torch.rand(B, N, 3, device=dev) * 2 - 1
    xyz2 = torch.rand(B, S, 3, device=dev) * 2 - 1
    feat = torch.randn(B, S, Cin, device=dev)
    g_out = torch.randn(B, N, mlp[-1], device=dev)
    ref = upp_layers.PointNetFeaturePropagation(Cin + 3, mlp, interpolate_neighbors=k).to(dev).train()
    res = {}
    for name, commute, dt in (("f64", False, torch.float64), ("concat", False, torch.float32), ("commuted", True, torch.float32)):
        m = upp_layers.PointNetFeaturePropagation(Cin + 3, mlp, interpolate_neighbors=k).to(dev).train()
        m.load_state_dict(ref.state_dict())
        m = m.to(dt)
        m.commute_first_layer = commute
        x = feat.to(dt).clone().requires_grad_(True)
        out = m(xyz1.to(dt), xyz2.to(dt), xyz1.to(dt), x)
        (out * g_out.to(dt)).sum().backward()
        res[name] = [t.detach().double() for t in (out, x.grad, m.mlp_convs[0].weight.grad, m.mlp_bns[0].bias.grad, m.mlp_bns[0].weight.grad,
                                                   m.mlp_convs[1].weight.grad, m.mlp_bns[0].running_mean, m.mlp_bns[0].running_var)]
    for r, a, b in zip(res["f64"], res["concat"], res["commuted"]):
        scale = r.abs().max().item()
        ea, eb = (a - r).abs(), (b - r).abs()
        assert eb.max().item() <= 4.0 * ea.max().item() + 1e-5 * scale, (eb.max().item(), ea.max().item(), scale)
        assert eb.mean().item() <= 3.0 * ea.mean().item() + 1e-6 * scale, (eb.mean().item(), ea.mean().item(), scale)


@pytest.mark.parametrize("n,C,W", [(65536, 1536, 3), (1000, 260, 3), (77, 8, 4), (5000, 512, 1)])
def test_weighted_column_sums_are_the_skinny_product(n, C, W):
    """upp_wcolsum_partials: x (n,W)^T . g (n,C) as W weighted column sums of g read once (the rank-3 xyz weight gradient of the
    segmentation head's first propagation layer); a column window of a wider matrix is served in place."""
    from upp_hip import ops
    gen = torch.Generator(device='cuda').manual_seed(n + C)
    wide = torch.randn(n, C + 8, device='cuda', generator=gen)
    g = wide[:, 4:4 + C]
    x = torch.randn(n, W, device='cuda', generator=gen)
    part = ops.wcolsum_partials(g, x)
    ref = x.double().t() @ g.double()
    np.testing.assert_allclose(part.double().sum(0).cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=3e-6 * ref.abs().max().item())


@pytest.mark.parametrize("B,N,keep,p", [(32, 1096, 972, 0.0), (3, 1624, 1459, 0.2), (2, 100, 100, 0.0), (1, 7, 3, 0.5)])
def test_rectify_tail_scores_nudges_and_selects_like_the_reference_formulation(B, N, keep, p):
    """upp_rectify_select: score head (Linear(32,64) -> ReLU -> dropout -> Linear(64,3)) + ||.||_2 + stable descending order + nudge + gather
    (reference models/Point_MAE_pretask_dev.py:491-512, models/Point_MAE_unify.py:553-559) against the torch ops it replaces; exact ties
    keep the lower index first, like torch's stable sort."""
    from upp_hip import ops
    gen = torch.Generator(device='cuda').manual_seed(B * N + keep)
    feat = torch.randn(B, N, 32, device='cuda', generator=gen)
    pts = torch.randn(B, N, 3, device='cuda', generator=gen)
    if N > 50:                                              # duplicated rows: equal scores
        feat[:, 40:45] = feat[:, 10:15]
    w0 = torch.randn(64, 32, device='cuda', generator=gen) * 0.2
    b0 = torch.randn(64, device='cuda', generator=gen) * 0.1
    w1 = torch.randn(3, 64, device='cuda', generator=gen) * 0.2
    b1 = torch.randn(3, device='cuda', generator=gen) * 0.1
    u = torch.rand(B * N, 64, device='cuda', generator=gen) if p > 0 else None
    out, pred, order, score = ops.rectify_select(feat, w0, b0, w1, b1, pts, keep, u, p, 1.5, 0.2, want_pred=True, want_order=True, want_score=True)
    h = torch.relu(feat.double() @ w0.double().t() + b0.double())
    if u is not None:
        h = torch.where(u.view(B, N, 64) >= p, h / (1.0 - p), torch.zeros_like(h))
    ref = (h @ w1.double().t() + b1.double()) * 1.5
    np.testing.assert_allclose(pred.double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=3e-6 * ref.abs().max().item())
    np.testing.assert_allclose(score.cpu().numpy(), torch.linalg.vector_norm(pred, dim=-1).cpu().numpy(), rtol=1e-6)
    want_order = torch.sort(score, dim=1, descending=True, stable=True)[1]
    assert torch.equal(order, want_order)
    moved = pts + pred * 0.2
    want = torch.gather(moved, 1, want_order[:, -keep:, None].expand(-1, -1, 3))
    assert torch.equal(out, want)


def test_rectify_tail_ranks_nan_scores_as_the_largest_like_argsort():
    """A NaN score compares false to everything; torch.argsort (the reference formulation) sorts NaN as the largest value and always
    returns a permutation.  The rank kernel does the same: no slot of `order` / `out` is left unwritten."""
    from upp_hip import ops
    gen = torch.Generator(device='cuda').manual_seed(5)
    B, N, keep = 2, 300, 250
    feat = torch.randn(B, N, 32, device='cuda', generator=gen)
    feat[0, 17, 0] = float('inf'); feat[0, 200, 0] = float('inf'); feat[1, 0, 0] = float('inf')
    pts = torch.randn(B, N, 3, device='cuda', generator=gen)
    # a score head that passes channel 0 through: score = relu(feat[..., 0]); an infinite channel gives pred = (inf, 0 * inf, 0 * inf) -> NaN norm
    w0, b0 = torch.zeros(64, 32, device='cuda'), torch.zeros(64, device='cuda')
    w1, b1 = torch.zeros(3, 64, device='cuda'), torch.zeros(3, device='cuda')
    w0[0, 0] = 1.0; w1[0, 0] = 1.0
    out, pred, order, score = ops.rectify_select(feat, w0, b0, w1, b1, pts, keep, None, 0.0, 1.0, 0.2, want_pred=True, want_order=True, want_score=True)
    assert bool(torch.isnan(score[0, 17])) and bool(torch.isnan(score[1, 0]))
    for b in range(B):
        assert sorted(order[b].tolist()) == list(range(N))                      # a permutation
    assert order[0, :2].tolist() == [17, 200] and order[1, 0].item() == 0          # NaN = the largest: first in descending order, ties in index order
    # (torch.sort on this ROCm build puts NaN LAST in a descending sort although its documentation ranks NaN above every number: the
    #  finite part is compared with torch, the NaN rows with the documented rule)
    for b, nn in ((0, 2), (1, 1)):
        finite = order[b, nn:]
        want = torch.sort(score[b][finite.sort()[0]], descending=True, stable=True)[1]
        assert torch.equal(finite, finite.sort()[0][want])


def test_batched_copy_moves_every_tensor_in_one_launch():
    """upp_copy_batched: the hand-over state of the pipelined step (mixed dtypes and sizes, odd byte counts)."""
    from upp_hip import ops
    gen = torch.Generator(device='cuda').manual_seed(1)
    srcs = [torch.randn(32, 64, 384, device='cuda', generator=gen), torch.randn(32, 64, 3, device='cuda', generator=gen),
            torch.randint(0, 1000, (8192,), device='cuda', generator=gen), torch.randint(0, 100, (2401,), device='cuda', dtype=torch.int32, generator=gen),
            torch.randn(7, device='cuda', generator=gen), torch.randint(0, 2, (33,), device='cuda', dtype=torch.uint8, generator=gen)]
    dsts = [torch.zeros_like(s) for s in srcs]
    ops.copy_batched(dsts, srcs)
    for d, s in zip(dsts, srcs):
        assert torch.equal(d, s)


@pytest.mark.parametrize("B,N,S,C,k", [(32, 32, 32, 384, 6), (8, 1096, 32, 32, 16), (2, 100, 7, 12, 16), (3, 5, 64, 256, 3)])
def test_interpolation_with_geometry_gradients_equals_the_reference_formulation(B, N, S, C, k):
    """HF.interp_geo (upp_sqdist_topk + upp_interp_fwd; upp_interp_bwd + upp_interp_geo_bwd) against the reference's autograd chain
    (models/Point_MAE_unify.py:22-48: square_distance, sort, 1/(d+eps) weights, index_points): output and the gradients w.r.t. the
    queries, the sources and the features."""
    from models import upp_layers as L
    gen = torch.Generator(device='cuda').manual_seed(B * N + S)
    x1 = torch.rand(B, N, 3, device='cuda', generator=gen).requires_grad_(True)
    x2 = torch.rand(B, S, 3, device='cuda', generator=gen).requires_grad_(True)
    f = torch.randn(B, S, C, device='cuda', generator=gen).requires_grad_(True)
    w = torch.randn(B, N, C, device='cuda', generator=gen)
    out = L._inverse_distance_interp(x1, x2, f, k, 1e-3)
    (out * w).sum().backward()
    got = [out.detach(), x1.grad.clone(), x2.grad.clone(), f.grad.clone()]
    x1.grad = x2.grad = f.grad = None
    L.FUSE_INTERP_GEO = False
    try:
        ref = L._inverse_distance_interp(x1, x2, f, k, 1e-3)
        (ref * w).sum().backward()
    finally:
        L.FUSE_INTERP_GEO = True
    want = [ref.detach(), x1.grad, x2.grad, f.grad]
    for a, b, name in zip(got, want, ("out", "g_xyz1", "g_xyz2", "g_feat")):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-5 * b.abs().max().item(), err_msg=name)


@pytest.mark.parametrize("B,N", [(32, 64), (32, 1076), (3, 8192), (1, 1), (5, 257), (70000, 16)])
@pytest.mark.parametrize("descending", [False, True])
def test_argsort_rows_is_torchs_stable_argsort(B, N, descending):
    """upp_argsort_rows (rank counting; replaces the device sorts of reference models/Point_MAE_pretask_dev.py:702-704 and
    models/Point_MAE.py:300-329): the permutation torch.argsort(..., stable=True) returns, bit for bit, with ties and repeated values."""
    from upp_hip import functional as HF
    g = torch.Generator(device='cuda').manual_seed(B * 131 + N)
    key = torch.randn(B, N, device='cuda', generator=g)
    key[:, ::3] = key[:, ::3].round()                           # many exact ties
    if N > 4:
        key[:, 2] = key[:, 4]
    got = HF.argsort_rows(key, descending=descending)
    want = torch.argsort(key, dim=-1, descending=descending, stable=True)
    assert got.dtype == torch.int64 and torch.equal(got, want)
    mask = torch.rand(B, N, device='cuda', generator=g) < 0.6   # the visible-first order of the masked auto-encoder
    assert torch.equal(HF.argsort_rows(mask), torch.argsort(mask.int(), dim=1, stable=True))


def test_argsort_rows_ranks_nan_above_inf_like_torch_and_stays_a_permutation():
    """Total order of the rank-counting kernel: NaN above +inf (last ascending, FIRST descending), -0 == +0 with ties by index -- the
    permutation torch.argsort(stable=True) returns for rows holding NaN, both infinities and both zeros."""
    from upp_hip import functional as HF
    nan, inf = float('nan'), float('inf')
    key = torch.tensor([[0.5, nan, -1.0, inf, nan, 0.5, -inf, 0.0, -0.0, inf],
                        [nan, nan, -0.0, 0.0, 1e-45, -1e-45, 3.4e38, -3.4e38, inf, -inf]], device='cuda')
    for desc in (False, True):
        got = HF.argsort_rows(key, descending=desc)
        assert torch.equal(got, torch.argsort(key, dim=-1, descending=desc, stable=True)), desc
        assert torch.equal(HF.argsort_rows(key, descending=desc, stable=False), got)      # (stable=False permits this order too)
    assert HF.argsort_rows(key).cpu().tolist()[0] == [6, 2, 7, 8, 0, 5, 3, 9, 1, 4]


@pytest.mark.parametrize("last", [None, 32])
@pytest.mark.parametrize("grad", [False, True])
def test_layer_norm_on_the_row_kernel_equals_nn_layer_norm(last, grad):
    """HF.layer_norm (upp_rowln_fwd with the identity / strip row map): the front-end's `self.norm` (reference models/Point_MAE_unify.py:588)
    and the decoder's norm over its last tokens (models/Point_MAE_pretask_dev.py:381), forward and -- for the pre-training recipe, where
    the norm is trainable -- gradients, against nn.LayerNorm."""
    from upp_hip import functional as HF
    torch.manual_seed(5)
    ln = torch.nn.LayerNorm(384).cuda()
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.uniform_(-0.3, 0.3)
    x = (torch.randn(8, 64, 384, device='cuda') * 2.0 + 0.5).requires_grad_(grad)
    if not grad:
        ln.requires_grad_(False)
    want = ln(x if last is None else x[:, -last:])
    got = HF.layer_norm(x, ln, last=last)
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)
    if grad:
        w = torch.randn_like(want)
        gx, gw, gb = torch.autograd.grad((want * w).sum(), [x, ln.weight, ln.bias])
        hx, hw, hb = torch.autograd.grad((got * w).sum(), [x, ln.weight, ln.bias])
        torch.testing.assert_close(hx, gx, rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(hw, gw, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(hb, gb, rtol=1e-4, atol=1e-4)
    x2 = torch.randn(100, 384, device='cuda')                     # 2-D rows
    torch.testing.assert_close(HF.layer_norm(x2, ln), ln(x2), rtol=2e-5, atol=2e-5)


def test_gather_rows_forward_and_sort_free_backward():
    """HF.gather_rows (the set abstraction's feature gather, reference models/Point_MAE_pretask_dev.py:409-413): rows bit-equal to
    advanced indexing, gradient equal to torch's index_put_(accumulate) up to the summation order, and the same bits on every call."""
    from upp_hip import functional as HF
    g = torch.Generator(device='cuda').manual_seed(9)
    B, S, C, M = 32, 64, 384, 512
    pts = torch.randn(B, S, C, device='cuda', generator=g, requires_grad=True)
    idx = torch.randint(0, S, (B, M), device='cuda', generator=g)
    idx[:, :40] = 3                                             # many repeats of one source row
    w = torch.randn(B, M, C, device='cuda', generator=g)
    got = HF.gather_rows(pts, idx)
    want = torch.gather(pts, 1, idx.unsqueeze(-1).expand(-1, -1, C))
    assert torch.equal(got, want)
    (gg,) = torch.autograd.grad((got * w).sum(), pts)
    (gw,) = torch.autograd.grad((want * w).sum(), pts)
    torch.testing.assert_close(gg, gw, rtol=1e-5, atol=1e-5)
    (gg2,) = torch.autograd.grad((HF.gather_rows(pts, idx) * w).sum(), pts)
    assert torch.equal(gg, gg2)
    with torch.no_grad():
        assert torch.equal(HF.gather_rows(pts, idx), want)


@pytest.mark.parametrize("shape", [(2048, 32, 384), (1024, 16, 256), (32, 32, 16, 12), (7, 1, 8), (3, 255, 4)])
def test_group_max_equals_torch_max_with_its_gradient(shape):
    """HF.group_max (upp_group_max_fwd / _bwd): the max-pool sites under autograd (reference models/Point_MAE_unify.py:205,221 patch
    embedding, models/Point_MAE_pretask_dev.py:413 set abstraction, :294 pooling) -- values bit-equal to torch.max over the group
    dimension, gradient equal to torch's scatter on tie-free data, the first maximal row on ties."""
    from upp_hip import functional as HF
    g = torch.Generator(device='cuda').manual_seed(sum(shape))
    x = torch.randn(*shape, device='cuda', generator=g).requires_grad_(True)
    got = HF.group_max(x)
    want = x.max(dim=-2)[0]
    assert torch.equal(got, want)
    w = torch.randn_like(want)
    (gg,) = torch.autograd.grad((got * w).sum(), x)
    (gw,) = torch.autograd.grad((want * w).sum(), x)
    assert torch.equal(gg, gw)
    with torch.no_grad():
        assert torch.equal(HF.group_max(x), want)
    t = torch.zeros(2, 5, 8, device='cuda', requires_grad=True)          # all ties: the first row of the group takes the gradient
    (gt,) = torch.autograd.grad(HF.group_max(t).sum(), t)
    assert float(gt[:, 0].sum()) == 16.0 and float(gt[:, 1:].abs().sum()) == 0.0
    # a NaN anywhere in a group reaches the pooled value (torch.max(dim) propagates it; a diverged activation must stay visible), and the
    # FIRST NaN row takes the gradient
    n = torch.randn(4, 9, 8, device='cuda', generator=g)
    n[0, 0, 1] = n[1, 3, 2] = n[1, 7, 2] = n[2, 8, 5] = float('nan')
    n.requires_grad_(True)
    got_n, want_n = HF.group_max(n), n.max(dim=-2)[0]
    assert torch.equal(torch.isnan(got_n), torch.isnan(want_n)) and torch.isnan(got_n).sum().item() == 3
    assert torch.equal(torch.nan_to_num(got_n, nan=7.0), torch.nan_to_num(want_n, nan=7.0))
    (gn,) = torch.autograd.grad(got_n, n, torch.ones_like(got_n))
    assert gn[1, 3, 2] == 1 and gn[1, 7, 2] == 0 and gn[0, 0, 1] == 1 and gn[2, 8, 5] == 1 and gn.sum() == got_n.numel()


def test_fan_out_sums_the_branch_gradients_in_one_launch():
    """HF.fan_out: n aliases of a tensor read by n consumers (`pos` in front of every Transformer block, reference
    models/Point_MAE_unify.py:288-294); the n gradients come back summed in branch order -- equal to autograd's own accumulation up to
    the order of additions, and ONE library launch."""
    from upp_hip import functional as HF
    g = torch.Generator(device='cuda').manual_seed(2)
    x = torch.randn(32, 65, 384, device='cuda', generator=g, requires_grad=True)
    ws = [torch.randn(32, 65, 384, device='cuda', generator=g) for _ in range(12)]
    parts = HF.fan_out(x, 12)
    assert len(parts) == 12 and all(torch.equal(p, x) for p in parts)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        (got,) = torch.autograd.grad(sum((p * w).sum() for p, w in zip(parts, ws)), x)
        torch.cuda.synchronize()
    want = torch.stack(ws).sum(0)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-5)
    names = [e.key for e in prof.key_averages()]
    assert sum('batched_sum' in k for k in names) >= 1
    assert HF.fan_out(x.detach(), 3)[0] is not None and len(HF.fan_out(x.detach(), 3)) == 3


@pytest.mark.parametrize("R,C,Cpad,with_bias", [(65536, 50, 52, True), (1000, 64, 64, False), (33, 7, 8, True), (4096, 50, 50, True)])
def test_log_softmax_rows_and_nll_mean_match_torch(R, C, Cpad, with_bias):
    """Round 6: the per-point tail of the segmentation head -- F.log_softmax over the first C columns of a (R, Cpad) matrix (+ class bias)
    and F.nll_loss (mean) -- on upp_logsoftmax_rows_* / upp_nll_mean_* (reference models/Point_MAE_unify_segment.py:433, :20-25): values,
    the gradient in the padded layout (pad columns exactly zero), the bias gradient, determinism."""
    g = torch.Generator(device='cuda').manual_seed(R + C)
    y0 = torch.randn(R, Cpad, device='cuda', generator=g) * 3
    b0 = torch.randn(C, device='cuda', generator=g) if with_bias else None
    target = torch.randint(0, C, (R,), device='cuda', generator=g)
    y = y0.clone().requires_grad_(True)
    b = b0.clone().requires_grad_(True) if with_bias else None
    logp = HF.log_softmax_rows(y, b, C)
    loss = HF.nll_mean(logp, target)
    assert type(loss.grad_fn).__name__ == '_NllMeanBackward' and type(logp.grad_fn).__name__ == '_LogSoftmaxRowsBackward'
    loss.backward()
    yt = y0.clone().requires_grad_(True)
    bt = b0.clone().requires_grad_(True) if with_bias else None
    logp_t = F.log_softmax(yt[:, :C] + (bt if with_bias else 0.0), dim=-1)
    loss_t = F.nll_loss(logp_t, target)
    loss_t.backward()
    close(logp, logp_t, rtol=1e-5, atol_scale=2e-6)
    np.testing.assert_allclose(loss.item(), loss_t.item(), rtol=2e-6)
    close(y.grad[:, :C], yt.grad[:, :C], rtol=1e-5, atol_scale=2e-6)
    assert Cpad == C or float(y.grad[:, C:].abs().max()) == 0.0
    if with_bias:
        close(b.grad, bt.grad, rtol=2e-5, atol_scale=5e-6)
    y2 = y0.clone().requires_grad_(True)
    loss2 = HF.nll_mean(HF.log_softmax_rows(y2, b0, C), target)
    loss2.backward()
    assert torch.equal(loss2, loss.detach()) and torch.equal(y2.grad, y.grad)                # deterministic
    # a general upstream gradient (not the NLL's one-hot): the log-softmax backward on its own
    w = torch.randn(R, C, device='cuda', generator=g)
    y3, yt3 = y0.clone().requires_grad_(True), y0.clone().requires_grad_(True)
    (HF.log_softmax_rows(y3, b0, C) * w).sum().backward()
    (F.log_softmax(yt3[:, :C] + (b0 if with_bias else 0.0), dim=-1) * w).sum().backward()
    close(y3.grad[:, :C], yt3.grad[:, :C], rtol=2e-5, atol_scale=5e-6)


@pytest.mark.parametrize("M,N,K", [(65536, 50, 256), (4096, 50, 256), (2048, 30, 128)])
def test_linear_with_an_output_width_that_is_no_multiple_of_4_needs_no_padded_weight(M, N, K):
    """Round 6: the 50-class layer of the segmentation head (reference models/Point_MAE_unify_segment.py:432) inside a step driver:
    HF._LinearPadN multiplies through the un-padded weight's persistent plane image into a ceil4(N)-column matrix, takes the data
    gradient through a persistent zero-padded W^T and sums only the first N rows of the weight-gradient tiles into the parameter's
    buffer -- against F.linear's autograd; no torch pad / fill / copy kernels."""
    from upp_hip import ops
    torch.manual_seed(M + N)
    w = torch.nn.Parameter(torch.randn(N, K, device='cuda') * 0.05)
    x = torch.randn(M, K, device='cuda', requires_grad=True)
    Np = (N + 3) // 4 * 4
    gy = torch.randn(M, Np, device='cuda')
    gy[:, N:] = 0.0                                                       # (what the log-softmax backward hands over)
    ref = torch.nn.functional.linear(x, w)
    gx_t, gw_t = torch.autograd.grad(ref, (x, w), gy[:, :N])
    buf = torch.full((N * K,), 0.125, device='cuda')
    was = ops.PLANES.managed, HF.TRANSPOSED.managed
    ops.PLANES.managed = HF.TRANSPOSED.managed = True
    try:
        assert HF.linear_pad_n_usable(x, w)
        warm = torch.zeros(N * K, device='cuda')                          # first use: the persistent plane image and padded W^T are made (one zero-fill)
        with HF.deferred_sums({w.data_ptr(): warm.view(N, K)}):
            torch.autograd.grad(HF._LinearPadN.apply(x, w), (x, w), gy, allow_unused=True)
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            HF.TRANSPOSED.refresh_trainable(); ops.PLANES.refresh_trainable()          # what a step driver does at the start of a step
            with HF.deferred_sums({w.data_ptr(): buf.view(N, K)}) as scope:
                y = HF._LinearPadN.apply(x, w)
                hx, hw = torch.autograd.grad(y, (x, w), gy, allow_unused=True)
            torch.cuda.synchronize()
    finally:
        ops.PLANES.managed, HF.TRANSPOSED.managed = was
    assert tuple(y.shape) == (M, Np) and hw is None and w.data_ptr() in scope.routed
    close(y[:, :N], ref, rtol=1e-5, atol_scale=2e-6)
    assert Np == N or float(y[:, N:].abs().max()) == 0.0
    close(hx, gx_t, rtol=1e-5, atol_scale=2e-6)
    close((buf - 0.125).view(N, K), gw_t, rtol=2e-5, atol_scale=5e-6)
    names = [e.key for e in prof.key_averages()]
    assert not any(("FillFunctor" in n_ or "reduce_kernel" in n_ or "direct_copy" in n_) for n_ in names), names


@pytest.mark.parametrize("B,P,pn", [(32, 1076, 1024), (3, 40, 7), (1, 2, 1)])
def test_noise_supervision_loss_in_two_launches_matches_the_torch_formula(B, P, pn):
    """Round 6: HF.noise_loss (upp_noise_loss_fwd / _bwd) against the reference's formula for the pre-task noise supervision
    (models/Point_MAE_pretask_dev.py:685-692): mean(norm(pred_noise - nv) ** 2) + mean(norm(pred_pure) ** 2), score = norm(pred),
    and the gradient w.r.t. pred -- including points that sit exactly on their target (norm's backward is guarded there; 2 (p - t) is 0)."""
    g = torch.Generator(device='cuda').manual_seed(B + P)
    pred0 = torch.randn(B, P, 3, device='cuda', generator=g) * 0.3
    nv = torch.randn(B, P - pn, 3, device='cuda', generator=g) * 0.2
    pred0[0, 0] = 0.0                                                    # a shape point with zero offset
    pred0[0, pn] = nv[0, 0]                                              # a noise point exactly on its target
    pred = pred0.clone().requires_grad_(True)
    loss, score = HF.noise_loss(pred, nv, pn)
    (loss * 1.7).backward()
    pt = pred0.clone().requires_grad_(True)
    positive = torch.mean(torch.norm(pt[:, pn:] - nv, 2, dim=-1, keepdim=True) ** 2)
    negative = torch.mean(torch.norm(pt[:, :pn], 2, dim=-1, keepdim=True) ** 2)
    ((positive + negative) * 1.7).backward()
    np.testing.assert_allclose(loss.item(), (positive + negative).item(), rtol=3e-6)
    close(score, torch.norm(pred0, p=2, dim=-1), rtol=1e-6, atol_scale=1e-7)
    close(pred.grad, pt.grad, rtol=1e-5, atol_scale=1e-6)
    assert not score.requires_grad
    l2, _ = HF.noise_loss(pred0.clone().requires_grad_(True), nv, pn)
    assert torch.equal(l2, loss.detach())                                 # deterministic


@pytest.mark.parametrize("B,N,M", [(32, 1096, 32), (4, 64, 64), (2, 5, 3)])
def test_fps_centres_gradient_in_one_launch(B, N, M):
    """Round 6: the backward of HF.fps_gather (upp_fps_gather_bwd): zero-fill + scatter of the centre gradients by the int32 FPS indices in one
    launch, against torch's index_add on the same indices."""
    g = torch.Generator(device='cuda').manual_seed(B * N + M)
    xyz = torch.randn(B, N, 3, device='cuda', generator=g).requires_grad_(True)
    centers, idx = HF.fps_gather(xyz, M)
    w = torch.randn(B, M, 3, device='cuda', generator=g)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        (gx,) = torch.autograd.grad(centers, xyz, w)
        torch.cuda.synchronize()
    want = torch.zeros(B, N, 3, device='cuda')
    want.scatter_add_(1, idx.long().unsqueeze(-1).expand(-1, -1, 3), w)
    assert torch.equal(gx, want)
    names = [e.key for e in prof.key_averages()]
    assert sum("fps_gather_bwd_kernel" in n_ for n_ in names) == 1 and not any("FillFunctor" in n_ or "direct_copy" in n_ for n_ in names), names


@pytest.mark.parametrize("K,H,D,x_grad", [(3, 128, 384, False), (3, 64, 384, True), (12, 32, 64, True)])
def test_position_mlp_with_gradients_keeps_gelu_in_the_kernels(K, H, D, x_grad):
    """Round 6: Linear(K, H) - GELU - Linear(H, D) with trainable weights (the position MLPs of stage 2 and the pre-task recipe, reference
    models/Point_MAE_pretask_dev.py:395-399) through upp_layers.mlp2: GELU and GELU' come out of the small-K launch, the second layer's
    data gradient is multiplied by GELU' in its epilogue -- values and every gradient against torch, and no torch gelu kernels."""
    torch.manual_seed(K + H)
    seq = torch.nn.Sequential(torch.nn.Linear(K, H), torch.nn.GELU(), torch.nn.Linear(H, D)).cuda()
    x0 = torch.randn(32, 33, K, device='cuda')
    w = torch.randn(32, 33, D, device='cuda')
    x = x0.clone().requires_grad_(x_grad)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        y = upp_layers.mlp2(seq, x)
        leaves = ([x] if x_grad else []) + list(seq.parameters())
        got = torch.autograd.grad((y * w).sum(), leaves)
        torch.cuda.synchronize()
    xt = x0.clone().requires_grad_(x_grad)
    yt = seq(xt)
    want = torch.autograd.grad((yt * w).sum(), ([xt] if x_grad else []) + list(seq.parameters()))
    close(y, yt, rtol=1e-5, atol_scale=2e-6)
    for a, b in zip(got, want):
        close(a, b, rtol=2e-5, atol_scale=5e-6)
    names = [e.key for e in prof.key_averages()]
    assert not any("gelu" in n_.lower() and "at::native" in n_ for n_ in names), names
