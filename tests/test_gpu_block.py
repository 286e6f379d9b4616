"""Fused Transformer-block kernels (csrc/block.hip) against the unfused torch composition that the golden
fixtures pin to the reference's Block / Attention.  Tolerance 1e-5 rel (north_star) + small abs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _seeded
from models import build_model_from_cfg, upp_layers
from upp_hip import functional as HF
from utils.config import builtin_cfg

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-5, atol_scale=2e-6):
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_scale * max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("L", [1, 15, 16, 17, 31, 35, 48, 64, 65, 75, 80, 81, 96, 97, 128, 129, 130, 131, 132, 133, 139, 143, 144, 145, 159, 160])
def test_attention_core_forward_backward(L):
    # attn_flash16.hip (L <= 96) / attn_long.hip (L <= 160) against the reference formulation (models/Point_MAE_pretask_dev.py:186-193)
    _attention_case(L)


def _attention_case(L):
    torch.manual_seed(L)
    B, H = 3, 6
    qkv = torch.randn(B, L, 3 * H * 64, device='cuda', requires_grad=True)
    w = torch.randn(B, L, H * 64, device='cuda')
    out = HF.attention(qkv, H, 0.125)
    (out * w).sum().backward()
    g = qkv.grad.clone(); qkv.grad = None
    q, k, v = qkv.view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (((q @ k.transpose(-2, -1)) * 0.125).softmax(-1) @ v).transpose(1, 2).reshape(B, L, H * 64)
    (ref * w).sum().backward()
    close(out, ref)
    close(g, qkv.grad, rtol=2e-5, atol_scale=5e-6)


@pytest.mark.parametrize("case", ["insert_cls", "insert", "strip_cls", "strip", "residual", "plain_noln"])
def test_rowln_forward_backward(case):
    torch.manual_seed(1)
    B, L, D, P = 4, 65, 384, 10
    dev = 'cuda'
    x = torch.randn(B, L + (P if case.startswith("strip") else 0), D, device=dev, requires_grad=True)
    pos = torch.randn_like(x).requires_grad_(True)
    prm = torch.randn(P, D, device=dev, requires_grad=True)
    gam = (1 + 0.1 * torch.randn(D, device=dev)).requires_grad_(True)
    bet = (0.1 * torch.randn(D, device=dev)).requires_grad_(True)
    u = torch.rand(B, device=dev)
    keep = 0.7

    def scale():
        return ((keep + u).floor() / keep).view(B, 1, 1)

    if case == "insert_cls":
        xo, h = HF.rowln(x, add=pos, prompts=prm, mode=HF.ROW_INSERT_CLS, P=P, gamma=gam, beta=bet)
        xp = x + pos
        rxo = torch.cat((xp[:, :1], prm.expand(B, -1, -1), xp[:, 1:]), 1)
    elif case == "insert":
        xo, h = HF.rowln(x, add=pos, prompts=prm, mode=HF.ROW_INSERT, P=P, gamma=gam, beta=bet)
        rxo = torch.cat((prm.expand(B, -1, -1), x + pos), 1)
    elif case == "strip_cls":
        y = torch.randn_like(x).requires_grad_(True)
        xo, h = HF.rowln(x, y=y, u=u, keep=keep, mode=HF.ROW_STRIP_CLS, P=P, gamma=gam, beta=bet)
        full = x + scale() * y
        rxo = torch.cat((full[:, :1], full[:, P + 1:]), 1)
    elif case == "strip":
        y = torch.randn_like(x).requires_grad_(True)
        xo, h = HF.rowln(x, y=y, u=u, keep=keep, mode=HF.ROW_STRIP, P=P, gamma=gam, beta=bet)
        rxo = (x + scale() * y)[:, P:]
    elif case == "residual":
        y = torch.randn_like(x).requires_grad_(True)
        xo, h = HF.rowln(x, y=y, u=u, keep=keep, gamma=gam, beta=bet)
        rxo = x + scale() * y
    else:
        y = torch.randn_like(x).requires_grad_(True)
        xo, h = HF.rowln(x, y=y)
        rxo = x + y
        assert h is None
    params = [t for t in (x, pos, prm, gam, bet) if True]
    w1, w2 = torch.randn_like(rxo), torch.randn_like(rxo)
    loss = (xo * w1).sum() + ((h * w2).sum() if h is not None else 0)
    grads = torch.autograd.grad(loss, [t for t in (x, pos, prm, gam, bet, locals().get('y')) if t is not None], allow_unused=True)
    rh = F.layer_norm(rxo, (D,), gam, bet, 1e-5) if h is not None else None
    rloss = (rxo * w1).sum() + ((rh * w2).sum() if rh is not None else 0)
    rgrads = torch.autograd.grad(rloss, [t for t in (x, pos, prm, gam, bet, locals().get('y')) if t is not None], allow_unused=True)
    close(xo, rxo)
    if h is not None:
        close(h, rh)
    for g, r in zip(grads, rgrads):
        assert (g is None) == (r is None)
        if g is not None:
            close(g, r, rtol=2e-5, atol_scale=5e-6)


@pytest.fixture(scope="module")
def model():
    m = build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)
    return _seeded.fill(m).eval().cuda()


def _block_case(model, name, B=4):
    g = torch.Generator(device='cuda').manual_seed(3)
    pts = _seeded.unit_ball_clouds(B, 1024, seed=2).cuda()
    with torch.no_grad():
        _, center = model.group_divider(pts)
        lvl2 = upp_layers.Group(32, 8)
        _, c2, i1, i2 = lvl2(center, require_index=True, gather_idx=False)
    if name in ("down0", "down7", "down3_gather"):
        L, blk = 65, model.blocks.blocks[{"down0": 0, "down7": 7, "down3_gather": 3}[name]]
        gi = name.endswith("gather")
        if gi:
            with torch.no_grad():
                _, c2, i1, i2 = lvl2(center, require_index=True, gather_idx=True)
        kw = dict(path='downstream', downstream_adapter=True, downstream_prompts=True, classification=True, center1=center,
                  center1_idx=i1, center2=c2, center2_idx=i2, gather_idx=gi, prompt_propagation_after=True, _prop_cache={})
    elif name == "rectify":
        L, blk, kw = 32, model.blocks.blocks[1], dict(path='rectify', rectify_adapter=True, rectify_prompts=True, rectify_depth=3)
    elif name == "pretask":
        L, blk, kw = 32, model.blocks.blocks[4], dict(path='pretask', pretask_adapter=True, pretask_prompts=True, pretask_depth=6)
    else:
        L, blk, kw = 64, model.MAE_decoder.blocks[2], dict(path='pretask', pretask_adapter=True)
    x = torch.randn(B, L, 384, device='cuda', generator=g)
    pos = torch.randn(B, L, 384, device='cuda', generator=g)
    return blk, x, pos, kw


@pytest.mark.parametrize("name", ["down0", "down3_gather", "down7", "rectify", "pretask", "decoder"])
def test_block_fused_equals_unfused_with_gradients(model, name):
    blk, x, pos, kw = _block_case(model, name)
    assert blk.fusable(x)
    params = [p for n, p in blk.named_parameters() if ('adapter' in n or 'prompts' in n or 'bnorm' in n or 'norm1' in n)]
    outs = []
    for fused in (True, False):
        xi, pi = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
        out = blk.forward_fused(xi, pi, **kw) if fused else blk(xi + pi, **kw)
        w = torch.linspace(-1, 1, out.numel(), device='cuda').view_as(out)
        grads = torch.autograd.grad((out * w).sum(), [xi, pi] + params, allow_unused=True)
        outs.append((out.detach(), grads))
    close(outs[0][0], outs[1][0], rtol=1e-5, atol_scale=3e-6)
    for g, r in zip(outs[0][1], outs[1][1]):
        assert (g is None) == (r is None)
        if g is not None:
            close(g, r, rtol=5e-5, atol_scale=1e-5)




def test_train_mode_drop_path_statistics(model):
    """Stochastic depth cannot be compared draw by draw; check the per-sample factor takes the two legal values."""
    blk, x, pos, kw = _block_case(model, "down7")
    blk.train()
    try:
        ref = blk.forward_fused(x, pos, **kw)
        assert torch.isfinite(ref).all()
        u = torch.tensor([0.0, 0.95, 0.5, 0.99], device='cuda')
        keep = 0.9
        y = torch.ones_like(x)
        xo, _ = HF.rowln(torch.zeros_like(x), y=y, u=u, keep=keep)
        got = xo[:, 0, 0].cpu().numpy()
        np.testing.assert_allclose(got, np.floor(keep + u.cpu().numpy()) / keep, rtol=1e-6)
    finally:
        blk.eval()


def test_flat_adamw_matches_torch_clip_and_adamw():
    """csrc/optim.hip against clip_grad_norm_(10) + torch.optim.AdamW (the reference's step tail)."""
    from upp_hip.train import FlatAdamW
    from utils.dist_utils import FlatGradAllReduce
    torch.manual_seed(0)
    shapes_nd, shapes_d = [(384,), (32,), (1, 1, 384)], [(32, 384), (384, 32), (10, 384)]
    mk = lambda shp: [torch.nn.Parameter(torch.randn(s, device='cuda') * 0.1) for s in shp]
    nd, d = mk(shapes_nd), mk(shapes_d)
    rnd, rd = [torch.nn.Parameter(p.detach().clone()) for p in nd], [torch.nn.Parameter(p.detach().clone()) for p in d]
    ref = torch.optim.AdamW([{'params': rnd, 'weight_decay': 0.}, {'params': rd, 'weight_decay': 0.05}], lr=5e-4)
    flat = FlatGradAllReduce(nd + d)
    opt = FlatAdamW(nd, d, flat.flat, lr=5e-4, max_norm=10.0)
    for it in range(5):
        scale = 50.0 if it % 2 == 0 else 0.01          # with and without active clipping
        for p, q, v in zip(nd + d, rnd + rd, flat.views):
            gr = torch.randn_like(p) * scale
            v.copy_(gr)
            q.grad = gr.clone()
        total = torch.nn.utils.clip_grad_norm_(rnd + rd, 10.0)
        ref.step()
        opt.step()
        np.testing.assert_allclose(opt.state[1].item(), total.item(), rtol=1e-5)
        for p, q in zip(nd + d, rnd + rd):
            close(p, q, rtol=2e-5, atol_scale=2e-6)
    assert opt.state[0].item() == 5
    # the torch.optim-style view follows set_lr / load_state_dict: a scheduler's sync must not push stale values back to the device
    groups = opt.param_groups
    sd = opt.state_dict()
    sd['param_groups'][-1]['weight_decay'] = 0.125
    sd['param_groups'][0]['lr'] = sd['param_groups'][-1]['lr'] = 2.5e-4
    opt.load_state_dict(sd)
    assert groups[-1]['weight_decay'] == 0.125 and groups[0]['lr'] == 2.5e-4
    for g_ in groups:
        g_['lr'] = 1e-4                                   # what a scheduler does
    opt.sync_param_groups()
    assert abs(opt.state[5].item() - 1e-4) < 1e-10 and abs(opt.state[6].item() - 0.125) < 1e-7      # lr / weight decay on the device
    opt.set_lr(3e-4)
    assert groups[0]['lr'] == groups[1]['lr'] == 3e-4 and groups[1]['weight_decay'] == 0.125


@pytest.mark.parametrize("R", [2400, 2080, 1120, 45])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_adapter_kernel_forward_backward(R, p):
    torch.manual_seed(R)
    D, H = 384, 32
    dev = 'cuda'
    ha = torch.randn(R, D, device=dev, requires_grad=True)
    x = torch.randn(R, D, device=dev, requires_grad=True)
    W1 = (torch.randn(H, D, device=dev) / D ** 0.5).requires_grad_(True)
    b1 = (0.1 * torch.randn(H, device=dev)).requires_grad_(True)
    W2 = (torch.randn(D, H, device=dev) / H ** 0.5).requires_grad_(True)
    b2 = (0.1 * torch.randn(D, device=dev)).requires_grad_(True)
    u = torch.rand(R, H, device=dev) if p > 0 else None
    w = torch.randn(R, D, device=dev)
    out = HF.adapter(ha, x, W1, b1, W2, b2, u, p, 0.7)
    grads = torch.autograd.grad((out * w).sum(), [ha, x, W1, b1, W2, b2])
    mask = ((u >= p).float() / (1 - p)) if p > 0 else 1.0
    ref = x + 0.7 * (torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(ha, W1, b1)) * mask, W2, b2))
    rgrads = torch.autograd.grad((ref * w).sum(), [ha, x, W1, b1, W2, b2])
    close(out, ref, rtol=1e-5, atol_scale=2e-6)
    for g, r in zip(grads, rgrads):
        close(g, r, rtol=5e-5, atol_scale=1e-5)


@pytest.mark.parametrize("B,Lin,P,mode,pd,with_y", [(32, 75, 10, 'strip_cls', 0.1, True), (32, 74, 10, 'strip', 0.0, True), (3, 33, 0, 'identity', 0.1, True),
                                                     (2, 21, 5, 'strip_cls', 0.0, False), (1, 16, 0, 'identity', 0.0, True), (32, 43, 10, 'strip', 0.1, True)])
def test_block_tail_in_one_launch_equals_row_kernel_plus_adapter(B, Lin, P, mode, pd, with_y):
    """upp_ln_adapter_fwd / _bwd (residual + strip + adapter LayerNorm + adapter, 16-row workgroups) against the two launches it
    replaces: rows, statistics bit-identical; the output and every gradient within f32 re-association of the two small products."""
    from upp_hip import ops
    torch.manual_seed(B * 100 + Lin)
    D, H, dev = 384, 32, 'cuda'
    m = {'strip_cls': HF.ROW_STRIP_CLS, 'strip': HF.ROW_STRIP, 'identity': HF.ROW_IDENTITY}[mode]
    Lout = Lin - P if m != HF.ROW_IDENTITY else Lin
    mk = lambda *s, sc=1.0: (sc * torch.randn(*s, device=dev)).requires_grad_(True)
    x, y = mk(B, Lin, D), (mk(B, Lin, D) if with_y else None)
    yb = 0.1 * torch.randn(D, device=dev) if with_y else None
    u = torch.rand(B, device=dev) if with_y else None
    ln = torch.nn.LayerNorm(D).to(dev)
    with torch.no_grad():
        ln.weight.add_(0.1 * torch.randn(D, device=dev)); ln.bias.add_(0.1 * torch.randn(D, device=dev))
    W1, b1, W2, b2 = mk(H, D, sc=D ** -0.5), mk(H, sc=0.1), mk(D, H, sc=H ** -0.5), mk(D, sc=0.1)
    ud = torch.rand(B * Lout, H, device=dev) if pd > 0 else None
    w = torch.randn(B, Lout, D, device=dev)
    leaves = [t for t in (x, y, ln.weight, ln.bias, W1, b1, W2, b2) if t is not None]
    out = HF.ln_adapter(x, y, yb, u, 0.9, m, P, ln, W1, b1, W2, b2, ud, pd, 0.7)
    grads = torch.autograd.grad((out * w).sum(), leaves)
    x4, ha = HF.rowln(x, y=y, ybias=yb, u=u, keep=0.9, mode=m, P=P, gamma=ln.weight, beta=ln.bias, eps=ln.eps)
    ref = HF.adapter(ha, x4, W1, b1, W2, b2, ud, pd, 0.7)
    rgrads = torch.autograd.grad((ref * w).sum(), leaves)
    # the saved rows / statistics of the fused forward are the row kernel's, bit for bit
    o2, xo, mean, rstd, s1 = ops.ln_adapter_fwd(x.detach(), None if y is None else y.detach(), yb, u, 0.9, m, P, ln.weight.detach(), ln.bias.detach(), ln.eps,
                                                W1.detach(), b1.detach(), W2.detach(), b2.detach(), ud, pd, 0.7, Lout)
    xo_r, h_r, mean_r, rstd_r = ops.rowln_fwd(x.detach(), None, None, m, P, None if y is None else y.detach(), u, 0.9, ln.weight.detach(), ln.bias.detach(), ln.eps, Lout, ybias=yb)
    assert torch.equal(xo, xo_r) and torch.equal(mean, mean_r) and torch.equal(rstd, rstd_r) and torch.equal(o2, out.detach())
    close(out, ref, rtol=1e-5, atol_scale=2e-6)
    for g, r in zip(grads, rgrads):
        close(g, r, rtol=5e-5, atol_scale=1e-5)
    # the one-launch backward (upp_ln_adapter_bwd_fused, 16-row workgroups) against the two-launch backward of the same Function
    HF.FUSE_TAIL_BACKWARD = False
    try:
        out2 = HF.ln_adapter(x, y, yb, u, 0.9, m, P, ln, W1, b1, W2, b2, ud, pd, 0.7)
        grads2 = torch.autograd.grad((out2 * w).sum(), leaves)
    finally:
        HF.FUSE_TAIL_BACKWARD = True
    for g, r in zip(grads, grads2):
        close(g, r, rtol=2e-5, atol_scale=4e-6)
    if m != HF.ROW_IDENTITY:      # the prompt rows the strip map dropped: exactly zero
        rows = slice(1, 1 + P) if m == HF.ROW_STRIP_CLS else slice(0, P)
        assert float(grads[0][:, rows].abs().max()) == 0.0
    # inside a deferred scope (the step driver's): the adapter weight gradients from per-row factors (upp_ln_adapter_bwd_factors +
    # upp_adapter_wgrad_batched, ONE launch for all queued blocks -- two here) against the per-workgroup partial matrices, and both
    # against float64 of the same products
    params = [ln.weight, ln.bias, W1, b1, W2, b2]
    got = {}
    for fac in (True, False):
        HF.ADAPTER_FACTORS = fac
        try:
            targets = {p_.data_ptr(): torch.zeros_like(p_) for p_ in params}
            with HF.deferred_sums(targets) as scope:
                o_a = HF.ln_adapter(x, y, yb, u, 0.9, m, P, ln, W1, b1, W2, b2, ud, pd, 0.7)
                o_b = HF.ln_adapter(x, y, yb, u, 0.9, m, P, ln, W1, b1, W2, b2, ud, pd, 0.7)          # a second "block" with the same parameters
                g_in = torch.autograd.grad((o_a * w).sum() + ((o_b * w).sum() * 0.5), [x], allow_unused=True)
            assert scope.routed == set(targets)
            got[fac] = ([targets[p_.data_ptr()] for p_ in params], g_in[0])
        finally:
            HF.ADAPTER_FACTORS = True
    for a_, b_, r_ in zip(got[True][0], got[False][0], grads[len(leaves) - 6:]):
        close(a_, b_, rtol=2e-5, atol_scale=4e-6)
        close(a_, 1.5 * r_, rtol=5e-5, atol_scale=1e-5)                # (autograd of one application, scaled: 1 + 0.5)
    assert torch.equal(got[True][1], got[False][1])                    # the data gradient does not depend on the form


# ------------------------------------------------------------------ fused propagation step (CSR + in-kernel BatchNorm)
@pytest.mark.parametrize("n,rows,seg", [(8192, 2400, None), (1024, 2400, None), (16384, 1024, (512, 32)), (5, 3, None), (4096, 15360, None)])
def test_csr_build_is_the_stable_inverse(n, rows, seg):
    from upp_hip import ops
    rng = np.random.default_rng(n + rows)
    if seg is None:
        keys = rng.integers(0, rows, n).astype(np.int32)
        keys[rng.integers(0, n, max(1, n // 50))] = rows + 7          # out-of-range keys are skipped
        keys[0] = -1
        absolute = keys.astype(np.int64)
        start, perm = ops.csr_build(torch.from_numpy(keys).cuda(), rows)
    else:
        seg_len, seg_rows = seg
        keys = rng.integers(0, seg_rows, n).astype(np.int32)
        absolute = (np.arange(n) // seg_len) * seg_rows + keys
        start, perm = ops.csr_build(torch.from_numpy(keys).cuda(), rows, seg_len=seg_len, seg_rows=seg_rows)
    valid = np.nonzero((absolute >= 0) & (absolute < rows))[0]
    order = valid[np.argsort(absolute[valid], kind='stable')]
    counts = np.bincount(absolute[valid], minlength=rows)
    want_start = np.concatenate([[0], np.cumsum(counts)])
    assert np.array_equal(start.cpu().numpy(), want_start)
    assert np.array_equal(perm.cpu().numpy()[:len(order)], order)


@pytest.mark.parametrize("training,drop", [(True, 0.1), (True, 0.0), (False, 0.0)])
@pytest.mark.parametrize("B,Lp,off", [(32, 75, 1), (3, 74, 0)])
def test_fused_propagation_matches_pool_batchnorm_interp(B, Lp, off, training, drop):
    torch.manual_seed(B + Lp)
    T, G2, D = 64, 32, 384
    dev = 'cuda'
    X = torch.randn(B, Lp, D, device=dev) * 0.7 + 0.3
    rows = B * Lp
    base = (torch.arange(B, device=dev) * Lp + off + (Lp - off - T)).view(B, 1)
    i1 = (base + torch.randint(0, T, (B, G2 * 8), device=dev)).reshape(-1).int().contiguous()
    i2 = (base + torch.stack([torch.randperm(T, device=dev)[:G2] for _ in range(B)])).reshape(-1).int().contiguous()
    i1[:16] = i1[0]                                                   # a heavily referenced row
    idx8 = torch.randint(0, G2, (B, T, 8), device=dev).int().contiguous()
    w8 = torch.rand(B, T, 8, device=dev)
    w8 = (w8 / w8.sum(-1, keepdim=True)).contiguous()
    u = torch.rand(B * G2, device=dev) if drop > 0 else None
    keep = 1.0 - drop
    index = HF.PropIndex(i1, i2, idx8, w8, rows)
    wgt = torch.linspace(-1, 1, X.numel(), device=dev).view_as(X)
    results = []
    for fused in (True, False):
        bn = torch.nn.BatchNorm1d(D).to(dev)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, D)); bn.bias.copy_(torch.linspace(-0.2, 0.2, D))
            bn.running_mean.copy_(torch.linspace(-0.1, 0.4, D)); bn.running_var.copy_(torch.linspace(0.6, 1.7, D))
        xi = X.clone().requires_grad_(True)
        if fused:
            out = HF.propagate(xi, bn, index, u, keep, training)
        else:
            pooled = HF.prop_pool(xi, i1, u, keep)
            lc = F.batch_norm(pooled, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)
            out = HF.prop_interp(xi, lc.view(B, G2, D), i2, idx8, w8)
        grads = torch.autograd.grad((out * wgt).sum(), [xi, bn.weight, bn.bias])
        results.append((out.detach(), grads, bn.running_mean.clone(), bn.running_var.clone()))
    (o1, g1, rm1, rv1), (o2, g2, rm2, rv2) = results
    close(o1, o2, rtol=1e-5, atol_scale=3e-6)
    close(rm1, rm2, rtol=1e-5, atol_scale=1e-6)
    close(rv1, rv2, rtol=1e-5, atol_scale=1e-6)
    for a, b in zip(g1, g2):
        close(a, b, rtol=3e-5, atol_scale=1e-5)
    # deterministic: a second fused run is bit-identical
    bn = torch.nn.BatchNorm1d(D).to(dev)
    xi = X.clone().requires_grad_(True)
    a = HF.propagate(xi, bn, index, u, keep, training)
    ga = torch.autograd.grad((a * wgt).sum(), [xi])[0]
    bn = torch.nn.BatchNorm1d(D).to(dev)
    xj = X.clone().requires_grad_(True)
    b = HF.propagate(xj, bn, index, u, keep, training)
    gb = torch.autograd.grad((b * wgt).sum(), [xj])[0]
    assert torch.equal(a, b) and torch.equal(ga, gb)


@pytest.mark.parametrize("is_cls,gather,training", [(True, False, True), (False, True, True), (True, False, False)])
def test_fused_propagation_gradients_reach_the_centres(is_cls, gather, training):
    """Stage 2 of the recipe: the centres require a gradient (they are functions of the prompted point cloud), and the interpolation weights
    1/(d+eps) depend on them.  The fused step (HF.prop_weights + upp_prop_w8_grad) against the reference's torch formulation
    (Block._propagate_prompts) in f32, arbitrated by the same formulation in f64: tokens, centre gradients, BatchNorm gradients."""
    from models import upp_layers as L
    torch.manual_seed(5 + is_cls)
    B, T, D, P = 8, 64, 384, 10
    dev = 'cuda'
    blk = L.Block(D, 6).to(dev).train(training)
    with torch.no_grad():
        blk.bnorm.weight.copy_(torch.linspace(0.5, 1.5, D)); blk.bnorm.bias.copy_(torch.linspace(-0.2, 0.2, D))
        blk.bnorm.running_mean.copy_(torch.linspace(-0.1, 0.4, D)); blk.bnorm.running_var.copy_(torch.linspace(0.6, 1.7, D))
    state = {k: v.clone() for k, v in blk.bnorm.state_dict().items()}
    c1 = (torch.rand(B, T, 3, device=dev) * 2 - 1)
    Lp = T + P + (1 if is_cls else 0)
    X = torch.randn(B, Lp, D, device=dev) * 0.7 + 0.3
    wgt = torch.linspace(-1, 1, X.numel(), device=dev).view_as(X)
    _, c2_const, i1, i2 = L.Group(T // 2, 8)(c1, require_index=True, gather_idx=gather)
    pick = (i2.view(B, -1) - (0 if gather else torch.arange(B, device=dev).view(B, 1) * T)).long()

    def run(kind, dtype):
        blk.bnorm.load_state_dict(state)
        blk.to(dtype)
        a = c1.to(dtype).requires_grad_(True)
        b = torch.gather(a, 1, pick.unsqueeze(-1).expand(-1, -1, 3)) * 1.0      # level-2 centres: a differentiable selection of the level-1 ones
        x = X.to(dtype).requires_grad_(True)
        kw = dict(center1=a, center1_idx=i1, center2=b, center2_idx=i2, gather_idx=gather, classification=is_cls, _prop_cache={})
        out = blk._propagate_fused(x, kw) if kind == 'fused' else blk._propagate_prompts(x, kw)[0]
        grads = torch.autograd.grad((out * wgt.to(dtype)).sum(), [x, a, blk.bnorm.weight, blk.bnorm.bias])
        blk.float()
        return [out.detach().double()] + [g.double() for g in grads]

    fused, plain, exact = run('fused', torch.float32), run('plain', torch.float32), run('plain', torch.float64)
    assert float(exact[2].abs().max()) > 1e-3                               # the centre gradient is not trivially zero
    for name, f, p_, e in zip(("tokens", "g_tokens", "g_centres", "g_bn_weight", "g_bn_bias"), fused, plain, exact):
        scale = float(e.abs().max())
        err_f, err_p = float((f - e).abs().max()) / scale, float((p_ - e).abs().max()) / scale
        assert err_f <= max(2.0 * err_p, 2e-6), (name, err_f, err_p)


def test_batched_sum_and_deferred_scope():
    from upp_hip import ops
    g = torch.Generator(device='cuda').manual_seed(1)
    shapes = [(75, 24992), (32, 384), (32, 3840), (1, 5), (33, 1), (17, 300)] * 13      # 78 jobs -> two launches
    jobs = []
    for n, width in shapes:
        part = torch.randn(n, width, device='cuda', generator=g)
        off = 0 if width < 10 else 3
        length = width - off - (0 if width < 10 else 2)
        acc = len(jobs) % 2 == 1
        dst = torch.full((length,), 2.0 if acc else float('nan'), device='cuda')
        jobs.append((part, off, n, length, part.stride(0), dst, acc))
    ops.batched_sum(jobs)
    for part, off, n, length, ld, dst, acc in jobs:
        ref = torch.full((length,), 2.0 if acc else 0.0, device='cuda')
        for i in range(n):
            ref += part[i, off:off + length]          # destination first, then ascending row order, as the kernel
        assert torch.equal(dst, ref)
    # many partial matrices into ONE destination (what the cls-position gradient of 12 blocks does)
    shared = torch.zeros(384, device='cuda')
    parts = [torch.randn(32, 65 * 384, device='cuda', generator=g) for _ in range(12)]
    ops.batched_sum([(p_, 0, 32, 384, p_.stride(0), shared, True) for p_ in parts])
    ref = torch.zeros(384, device='cuda')
    for p_ in parts:
        for i in range(32):
            ref += p_[i, :384]
    assert torch.equal(shared, ref)
    # a deferred scope accumulates the same parameter gradients into the registered buffers
    torch.manual_seed(0)
    x = torch.randn(32, 65, 384, device='cuda')
    ln = torch.nn.LayerNorm(384).cuda()
    prompts = torch.randn(10, 384, device='cuda', requires_grad=True)
    wgt = None
    def loss_of(xi):
        xo, h = HF.rowln(xi, prompts=prompts, mode=HF.ROW_INSERT_CLS, P=10, gamma=ln.weight, beta=ln.bias, eps=ln.eps)
        return (h * torch.linspace(-1, 1, h.numel(), device='cuda').view_as(h)).sum() + xo.square().sum()
    params = [prompts, ln.weight, ln.bias]
    xi = x.clone().requires_grad_(True)
    want = torch.autograd.grad(loss_of(xi), [xi] + params)
    targets = {p.data_ptr(): torch.zeros_like(p) for p in params}
    xi = x.clone().requires_grad_(True)
    with HF.deferred_sums(targets) as scope:
        got = torch.autograd.grad(loss_of(xi), [xi] + params, allow_unused=True)
    assert got[1] is None and got[2] is None and got[3] is None and scope.routed == set(targets)
    close(got[0], want[0], rtol=1e-6, atol_scale=1e-6)
    for p_, w in zip(params, want[1:]):      # same partials, summed sequentially (kernel) vs torch's tree order: f32 reassociation
        close(targets[p_.data_ptr()], w, rtol=1e-4, atol_scale=2e-4)


def test_the_pipelines_side_stream_runs_beside_the_current_one():
    """torch hands out streams from a pool and the runtime maps them onto a few hardware queues: a side stream that shares the current
    stream's queue serialises the two-stream step (measured: 6.05 instead of 4.18 ms, by nothing but the number of streams created
    before).  PipelinedTrainStep therefore probes: whatever the pool hands out next, the stream it keeps runs beside the current one."""
    from upp_hip.train import _runs_beside, _concurrent_stream
    dev = torch.device('cuda', torch.cuda.current_device())
    cur = torch.cuda.current_stream(dev)
    assert not _runs_beside(cur, cur)                      # (one queue: one after the other -- the probe can tell)
    for shift in range(6):
        for _ in range(shift):
            torch.cuda.Stream(device=dev)                  # move the pool on
        s = _concurrent_stream(dev)
        assert s != cur and _runs_beside(cur, s)


def test_train_step_flat_gradients_equal_plain_autograd():
    """TrainStep (deferred sums, flat buffer, HIP graph) against loss.backward() on the same weights and batch."""
    from upp_hip.train import TrainStep, freeze_for_peft
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda().eval()
    freeze_for_peft(m)
    pts = _seeded.noisy_clouds(4, 1024, seed=9).cuda()
    labels = torch.tensor([1, 5, 9, 30], device='cuda')
    kw = dict(completion_prompt=True, denoise=True, point_num=1024)

    def reference():
        for p in m.parameters():
            p.grad = None
        loss, _ = m.get_loss_acc(m(pts, **kw), labels)
        loss.backward()
        out = {n: p.grad.clone() for n, p in m.named_parameters() if p.requires_grad and p.grad is not None}
        for p in m.parameters():
            p.grad = None
        return out

    names = {id(p): n for n, p in m.named_parameters()}
    for use_graph in (False, True):
        ts = TrainStep(m, tuple(pts.shape), use_graph=use_graph)
        ts.pts.copy_(pts); ts.labels.copy_(labels)
        if use_graph:
            ts._capture()                      # warm-up steps move the weights: take the reference afterwards
        want = reference()
        if use_graph:
            ts._g_fb.replay()
        else:
            ts._forward_backward()
        torch.cuda.synchronize()
        checked = 0
        for p, v in zip(ts.trainable, ts.flat.views):
            n = names[id(p)]
            if n in want:
                close(v, want[n], rtol=2e-4, atol_scale=2e-4)
                checked += 1
            else:
                assert float(v.abs().max()) == 0.0
        assert checked > 100


@pytest.mark.parametrize("gather_idx", [False, True])
@pytest.mark.parametrize("B,Lp,off,G2", [(32, 75, 1, 32), (3, 74, 0, 32), (2, 139, 1, 64)])
def test_prop_index_kernel_matches_the_torch_formulation(B, Lp, off, G2, gather_idx):
    from upp_hip import ops
    T = 2 * G2
    pts = _seeded.unit_ball_clouds(B, 1024, seed=B + Lp).cuda()
    with torch.no_grad():
        _, c1 = upp_layers.Group(T, 8)(pts)
        _, c2, i1, i2 = upp_layers.Group(G2, 8)(c1.contiguous(), require_index=True, gather_idx=gather_idx)
        want = upp_layers._prop_lists_torch(c1, c2, i1, i2, gather_idx, B, Lp, off)
    got = ops.prop_index(c1.contiguous(), c2.contiguous(), i1.contiguous(), i2.contiguous(), gather_idx, Lp, off, 1e-3)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # neighbour sets: identical wherever the 8th and 9th distances are separated by more than rounding noise
    d = upp_layers.square_distance(c1, c2).sort(dim=-1)[0]
    clear = (d[:, :, 8] - d[:, :, 7]) > 1e-6
    same = (got[2].long().sort(dim=-1)[0] == want[2].long().sort(dim=-1)[0]).all(-1)
    assert bool(same[clear].all()) and float(clear.float().mean()) > 0.95
    order_ok = (got[2] == want[2]).all(-1) | ~clear
    rows = order_ok & ((d[:, :, 1:8] - d[:, :, 0:7]).min(-1)[0] > 1e-6)
    close(got[3][rows], want[3][rows], rtol=2e-4, atol_scale=1e-6)
    close(got[3].sum(-1), torch.ones_like(got[3].sum(-1)), rtol=1e-6, atol_scale=1e-6)


def test_pipelined_train_step_equals_the_sequential_order():
    """front(k+1) || back(k) on two streams == the same calls run one after the other on the real BatchNorm buffers."""
    from upp_hip.train import TrainStep, PipelinedTrainStep, freeze_for_peft

    def make():
        m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda().train()
        for mod in m.modules():               # identical random streams are impossible across the two orders: no dropout
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if hasattr(mod, 'drop_prob'):
                mod.drop_prob = 0.0
        freeze_for_peft(m)
        return m

    B, steps = 4, 4
    raws = [_seeded.noisy_clouds(B, 1024, seed=40 + k).cuda() for k in range(steps)]
    labels = [torch.tensor([k, 2 * k + 1, 7, 39 - k], device='cuda') for k in range(steps)]
    kw = dict(completion_prompt=True, denoise=True, point_num=1024)

    m_ref = make()
    ref = TrainStep(m_ref, tuple(raws[0].shape), use_graph=False, forward_kwargs=kw)
    kw_back = dict(kw, completion_prompt=False, denoise=False)
    losses_ref = []
    with torch.no_grad():
        prev = m_ref.prompt_tokens(raws[0], True, True, 1024)
    for k in range(1, steps):
        with torch.no_grad():
            nxt = m_ref.prompt_tokens(raws[k], True, True, 1024)
        ref._forward_backward(prev, labels[k - 1], kw_back)
        ref._update()
        losses_ref.append(float(ref.loss))
        prev = nxt
    ref._forward_backward(prev, labels[steps - 1], kw_back)
    ref._update()
    losses_ref.append(float(ref.loss))

    m_pipe = make()
    pipe = PipelinedTrainStep(m_pipe, tuple(raws[0].shape), forward_kwargs=kw)
    before = {k: v.clone() for k, v in m_pipe.state_dict().items()}
    pipe._capture()                            # eager warm-up steps + capture: must leave model and optimizer as they were
    for k, v in m_pipe.state_dict().items():
        assert torch.equal(v, before[k]), "capture warm-up changed %s" % k
    assert float(pipe.opt.state[0]) == 0.0 and float(pipe.opt.m.abs().max()) == 0.0 and float(pipe.opt.v.abs().max()) == 0.0
    losses = []
    for k in range(steps):
        pipe.step(raws[k], labels[k])
        if k > 0:
            losses.append(float(pipe.loss))
    pipe.flush()
    losses.append(float(pipe.loss))
    np.testing.assert_allclose(losses, losses_ref, rtol=2e-4)
    sd_ref, sd = m_ref.state_dict(), m_pipe.state_dict()
    for kname in sd_ref:
        if 'num_batches_tracked' in kname:
            continue
        close(sd[kname], sd_ref[kname], rtol=5e-4, atol_scale=5e-5)
    enc_counters = [kname for kname in sd_ref if kname.startswith('encoder.') and 'num_batches_tracked' in kname]
    assert enc_counters and all(int(sd[kname]) == 3 * steps for kname in enc_counters)


def test_pipelined_step_merges_batchnorm_statistics_shared_by_both_halves():
    """A back-end that updates a BatchNorm the front-end also updates: shadow buffers + merge == the sequential order."""
    import copy
    import torch.nn as nn
    from upp_hip.train import PipelinedTrainStep

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = nn.Sequential(nn.Linear(3, 8), nn.BatchNorm1d(8))
            self.cls_head = nn.Linear(8, 4)
            for p in self.encoder.parameters():
                p.requires_grad_(False)

        def prompt_tokens(self, pts, completion_prompt=True, denoise=True, point_num=16):
            B, N, _ = pts.shape
            return self.encoder(pts.reshape(-1, 3)).view(B, N, 8), pts[:, :1].contiguous()

        def forward_tokens(self, tok, cen):
            y = self.encoder[1](tok.reshape(-1, 8) * 2.0 + cen.mean()).view_as(tok)      # the shared BatchNorm again
            return self.cls_head(y.mean(1))

        def forward(self, pts, **kw):
            return self.forward_tokens(*self.prompt_tokens(pts))

        def get_loss_acc(self, ret, gt):
            return nn.functional.cross_entropy(ret, gt), (ret.argmax(-1) == gt).float().mean() * 100

    torch.manual_seed(0)
    base = Toy().cuda().train()
    steps, B = 5, 6
    raws = [torch.randn(B, 16, 3, device='cuda') * (1 + 0.3 * k) + 0.2 * k for k in range(steps)]
    labels = [torch.randint(0, 4, (B,), device='cuda') for _ in range(steps)]
    # sequential order front(0), [front(k), back(k-1), sgd-free: gradients only], back(last) on the real buffers
    ref = copy.deepcopy(base)
    with torch.no_grad():
        prev = ref.prompt_tokens(raws[0])
    for k in range(1, steps):
        with torch.no_grad():
            nxt = ref.prompt_tokens(raws[k])
        ref.forward_tokens(*prev)
        prev = nxt
    ref.forward_tokens(*prev)
    pipe_model = copy.deepcopy(base)
    ts = PipelinedTrainStep(pipe_model, (B, 16, 3), forward_kwargs=dict(completion_prompt=True, denoise=True, point_num=16), lr=0.0)
    ts._capture()
    assert ts._merge
    pipe_model.load_state_dict(base.state_dict())         # undo the capture warm-up (lr = 0: only the statistics moved)
    for k in range(steps):
        ts.step(raws[k], labels[k])
    ts.flush()
    bn_ref, bn = ref.encoder[1], pipe_model.encoder[1]
    close(bn.running_mean, bn_ref.running_mean, rtol=1e-5, atol_scale=1e-6)
    close(bn.running_var, bn_ref.running_var, rtol=1e-5, atol_scale=1e-6)
    assert int(bn.num_batches_tracked) == int(bn_ref.num_batches_tracked) == 2 * steps


@pytest.mark.parametrize("B,L,D", [(32, 65, 384), (3, 2, 384), (5, 130, 96), (2, 40, 500)])
def test_cls_pool_matches_layernorm_max_concat(B, L, D):
    torch.manual_seed(B + L)
    x = torch.randn(B, L, D, device='cuda') * 1.3 + 0.2
    ln = torch.nn.LayerNorm(D).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.linspace(0.5, 1.5, D)); ln.bias.copy_(torch.linspace(-0.3, 0.3, D))
    for p_ in ln.parameters():
        p_.requires_grad_(False)
    w = torch.linspace(-1, 1, B * 2 * D, device='cuda').view(B, 2 * D)
    outs = []
    for fused in (True, False):
        xi = x.clone().requires_grad_(True)
        if fused:
            feat = HF.cls_pool(xi, ln)
        else:
            h = ln(xi)
            feat = torch.cat([h[:, 0], h[:, 1:].max(1)[0]], dim=-1)
        outs.append((feat.detach(), torch.autograd.grad((feat * w).sum(), xi)[0]))
    close(outs[0][0], outs[1][0], rtol=1e-5, atol_scale=2e-6)
    close(outs[0][1], outs[1][1], rtol=2e-5, atol_scale=5e-6)


@pytest.mark.parametrize("B,C", [(32, 40), (4, 40), (7, 3), (64, 500), (1, 16)])
def test_cross_entropy_acc_matches_torch(B, C):
    torch.manual_seed(B * C)
    logits = (torch.randn(B, C, device='cuda') * 3).requires_grad_(True)
    labels = torch.randint(0, C, (B,), device='cuda')
    with torch.no_grad():
        logits[0, labels[0]] += 20.0                      # at least one correct prediction
    loss, acc = HF.cross_entropy_acc(logits, labels)
    (g,) = torch.autograd.grad(loss * 1.7, logits)
    ref_logits = logits.detach().clone().requires_grad_(True)
    ref = F.cross_entropy(ref_logits, labels)
    (rg,) = torch.autograd.grad(ref * 1.7, ref_logits)
    close(loss, ref, rtol=1e-6, atol_scale=1e-6)
    close(g, rg, rtol=1e-5, atol_scale=1e-5)      # softmax - 1 at a confidently correct label is pure cancellation noise
    want_acc = (logits.argmax(-1) == labels).sum() / float(B) * 100
    assert abs(float(acc) - float(want_acc)) < 1e-4 and not acc.requires_grad


@pytest.mark.parametrize("R,C", [(32, 256), (4, 256), (7, 40), (130, 100)])
@pytest.mark.parametrize("training,p", [(True, 0.5), (True, 0.0), (False, 0.0)])
def test_bn_relu_drop_matches_torch_ops(R, C, training, p):
    torch.manual_seed(R * C)
    z = torch.randn(R, C, device='cuda') * 1.5 + 0.3
    u = torch.rand(R, C, device='cuda') if p > 0 else None
    w = torch.linspace(-1, 1, R * C, device='cuda').view(R, C)
    outs = []
    for fused in (True, False):
        bn = torch.nn.BatchNorm1d(C).cuda()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, C)); bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
            bn.running_mean.copy_(torch.linspace(-0.2, 0.2, C)); bn.running_var.copy_(torch.linspace(0.6, 1.5, C))
        zi = z.clone().requires_grad_(True)
        if fused:
            a = HF.bn_relu_drop(zi, bn, u, p, training)
        else:
            a = F.relu(F.batch_norm(zi, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps))
            if p > 0:
                a = a * (u >= p).float() / (1.0 - p)
        grads = torch.autograd.grad((a * w).sum(), [zi, bn.weight, bn.bias])
        outs.append((a.detach(), grads, bn.running_mean.clone(), bn.running_var.clone()))
    close(outs[0][0], outs[1][0], rtol=1e-5, atol_scale=2e-6)
    close(outs[0][2], outs[1][2], rtol=1e-5, atol_scale=1e-6)
    close(outs[0][3], outs[1][3], rtol=1e-5, atol_scale=1e-6)
    for a_, b_ in zip(outs[0][1], outs[1][1]):
        close(a_, b_, rtol=3e-5, atol_scale=1e-5)


def test_rowln_ybias_and_bias_gelu_match_the_biased_ops():
    torch.manual_seed(5)
    B, L, D = 8, 75, 384
    x = torch.randn(B, L, D, device='cuda'); y = torch.randn(B, L, D, device='cuda'); bias = torch.randn(D, device='cuda') * 0.3
    u = torch.rand(B, device='cuda')
    g1, b1 = torch.rand(D, device='cuda') + 0.5, torch.randn(D, device='cuda') * 0.1
    outs = []
    for fused in (True, False):
        xi, yi = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        if fused:
            xo, h = HF.rowln(xi, y=yi, ybias=bias, u=u, keep=0.9, gamma=g1, beta=b1)
        else:
            xo, h = HF.rowln(xi, y=yi + bias, u=u, keep=0.9, gamma=g1, beta=b1)
        grads = torch.autograd.grad((h * torch.linspace(-1, 1, h.numel(), device='cuda').view_as(h)).sum() + xo.sum(), [xi, yi])
        outs.append((xo.detach(), h.detach(), grads))
    close(outs[0][0], outs[1][0], rtol=1e-6, atol_scale=1e-6)
    close(outs[0][1], outs[1][1], rtol=1e-5, atol_scale=2e-6)
    for a, b in zip(outs[0][2], outs[1][2]):
        close(a, b, rtol=1e-5, atol_scale=2e-6)
    z = (torch.randn(2400, 1536, device='cuda') * 1.5).requires_grad_(True)
    bz = torch.randn(1536, device='cuda') * 0.5
    w = torch.linspace(-1, 1, z.numel(), device='cuda').view_as(z)
    got = HF.bias_gelu(z, bz)
    (gg,) = torch.autograd.grad((got * w).sum(), z)
    zr = z.detach().clone().requires_grad_(True)
    ref = F.gelu(zr + bz)
    (gr,) = torch.autograd.grad((ref * w).sum(), zr)
    close(got, ref, rtol=1e-6, atol_scale=1e-6)
    close(gg, gr, rtol=1e-5, atol_scale=1e-6)


def test_pipelined_step_with_a_recipe_back_end_equals_the_sequential_order():
    """PipelinedTrainStep(front_fn, back_fn, extras) on the part-segmentation recipe: front(k+1) || back(k) on two streams
    gives the losses and parameters of the same calls run one after the other."""
    from upp_hip.train import TrainStep, PipelinedTrainStep, freeze_for_peft
    keys = ['downstream_adapter', 'downstream_prompts', 'bnorm', 'label_conv', 'propagation_0', 'seg_head']

    def make():
        m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_shapenetpart_seg').model)).cuda().train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if hasattr(mod, 'drop_prob'):
                mod.drop_prob = 0.0
        freeze_for_peft(m, keys)
        return m

    B, steps = 2, 3
    raws = [torch.cat([_seeded.noisy_clouds(B, 1536, seed=60 + k), _seeded.unit_ball_clouds(B, 16, seed=70 + k) * 1.01], 1).contiguous().cuda()
            for k in range(steps)]
    lpts = [_seeded.unit_ball_clouds(B, 2048, seed=80 + k).cuda() for k in range(steps)]
    onehot = torch.zeros(B, 16, device='cuda')
    onehot[torch.arange(B), torch.arange(B) % 16] = 1
    g = torch.Generator(device='cuda').manual_seed(3)
    targets = [torch.randint(0, 50, (B * 2048,), device='cuda', generator=g) for _ in range(steps)]

    def front_fn(m, x):
        return m.prompt_tokens(x, True, True, 1536)

    def back_fn(m, state, onehot, lp, target):
        loss = m.get_loss(m.forward_tokens(state, onehot, lp).reshape(-1, 50), target)
        return loss, loss.detach()

    m_ref = make()
    # Adam turns rounding-level gradients (e.g. of a conv bias in front of a BatchNorm) into +-lr updates, so an eager and a
    # graph-replayed run drift apart at any learning rate.  The ORDER of the calls and the hand-over of state and extras is what
    # is tested: lr = 0, and the running statistics of the trainable heads' BatchNorms (a momentum-weighted, order-sensitive
    # function of every batch's label points and tokens) must agree next to the losses.
    lr = 0.0
    ref = TrainStep(m_ref, tuple(raws[0].shape), use_graph=False, loss_fn=lambda m: None, inputs=[], lr=lr)
    losses_ref = []
    for k in range(steps):
        with torch.no_grad():
            state = front_fn(m_ref, raws[k])
        ref.loss_fn = lambda m, state=state, k=k: back_fn(m, tuple(state), onehot, lpts[k], targets[k])
        ref._forward_backward()
        ref._update()
        losses_ref.append(float(ref.loss))

    m_pipe = make()
    pipe = PipelinedTrainStep(m_pipe, tuple(raws[0].shape), forward_kwargs=dict(completion_prompt=True, denoise=True, point_num=1536),
                              front_fn=front_fn, back_fn=back_fn, extras=[onehot, lpts[0], targets[0]], back_end_keys=tuple(keys), lr=lr)
    before = {k: v.clone() for k, v in m_pipe.state_dict().items()}
    pipe._capture()
    for k, v in m_pipe.state_dict().items():       # the capture warm-up leaves no trace (front-end BatchNorms included)
        assert torch.equal(v, before[k]), "capture warm-up changed %s" % k
    losses = []
    for k in range(steps):
        pipe.step(raws[k], extras=[onehot, lpts[k], targets[k]])
        if k > 0:
            losses.append(float(pipe.loss))
    pipe.flush()
    losses.append(float(pipe.loss))
    np.testing.assert_allclose(losses, losses_ref, rtol=2e-5)
    sd_ref, sd = m_ref.state_dict(), m_pipe.state_dict()
    assert any('running_var' in kname and kname.startswith('propagation_0') for kname in sd_ref)
    for kname in sd_ref:
        if 'num_batches_tracked' in kname:
            assert torch.equal(sd[kname], sd_ref[kname]), kname
            continue
        close(sd[kname], sd_ref[kname], rtol=1e-4, atol_scale=1e-5)


@pytest.mark.parametrize("rows,K,N", [(832, 384, 1536), (2048, 1536, 384), (300, 64, 40)])
def test_linear_with_a_deferred_bias_gradient_matches_autograd(rows, K, N):
    """HF.linear: data / weight gradients by the two GEMMs, the bias gradient by the batched column sum of the scope
    (tall-job kernel above 512 rows), accumulated into the registered buffer -- against torch's addmm backward."""
    torch.manual_seed(rows + N)
    x = torch.randn(4, rows // 4, K, device='cuda', requires_grad=True)
    w = (torch.randn(N, K, device='cuda') * 0.05).requires_grad_(True)
    b = torch.randn(N, device='cuda', requires_grad=True)
    g = torch.randn(4, rows // 4, N, device='cuda')
    ref_out = torch.nn.functional.linear(x, w, b)
    gx, gw, gb = torch.autograd.grad(ref_out, (x, w, b), g)
    buf = torch.full((N,), 0.5, device='cuda')                     # accumulated into: starts from a known value
    with HF.deferred_sums({b.data_ptr(): buf}) as scope:
        out = HF.linear(x, w, b)
        hx, hw, hb = torch.autograd.grad(out, (x, w, b), g, allow_unused=True)
    assert hb is None and b.data_ptr() in scope.routed
    close(out, ref_out)
    close(hx, gx, rtol=1e-5)
    close(hw, gw, rtol=1e-5)
    close(buf - 0.5, gb, rtol=1e-5, atol_scale=1e-5)
    # outside a scope: summed at once
    out = HF.linear(x, w, b)
    _, _, hb = torch.autograd.grad(out, (x, w, b), g)
    close(hb, gb, rtol=1e-5, atol_scale=1e-5)


@pytest.mark.parametrize("N,Kt,a", [(1536, 1155, 3), (512, 3456, 1024), (64, 200, 72)])
def test_column_windows_of_a_weight_train_without_copies(N, Kt, a):
    """Round 6: a Linear layer applied to COLUMN RANGES of one wider weight (the segmentation head's first layer on [per-point 1024 |
    per-sample 2432] columns of a (512, 3456, 1) Conv1d weight, the feature propagation's first layer on [xyz 3 | features 1152] of a
    (1536, 1155, 1) one: reference models/Point_MAE_unify_segment.py:424-433, models/Point_MAE_pretask_dev.py:425-473).  Inside a step
    driver (managed plane images + deferred sums) each range is multiplied where it lies -- also when its rows start off a 16-byte
    boundary -- and its weight gradient lands in that range of the parameter's gradient buffer: autograd sees None for the weight, no
    zero-fill / strided copy / add of the slice's backward, no torch reduction.  Against F.linear's autograd."""
    from upp_hip import ops
    torch.manual_seed(N + a)
    P = torch.nn.Parameter(torch.randn(N, Kt, 1, device='cuda') * 0.05)            # a Conv1d weight: trailing singleton dimension
    w = P.squeeze(-1)
    rows = 4096
    x1 = torch.randn(rows, a, device='cuda', requires_grad=True)
    x2 = torch.randn(rows, Kt - a, device='cuda', requires_grad=True)
    g = torch.randn(rows, N, device='cuda')
    ref = torch.nn.functional.linear(x1, w[:, :a]) + torch.nn.functional.linear(x2, w[:, a:])
    gx1, gx2, gP = torch.autograd.grad(ref, (x1, x2, P), g)
    buf = torch.full((N * Kt,), 0.25, device='cuda')
    was = ops.PLANES.managed
    ops.PLANES.managed = True
    try:
        ops.PLANES.refresh_trainable()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            with HF.deferred_sums({P.data_ptr(): buf.view(N, Kt, 1)}) as scope:
                out = HF.linear(x1, w[:, :a]) + HF.linear(x2, w[:, a:])
                ops.PLANES.refresh_trainable()                                       # (windows first seen in this pass get their images)
                hx1, hx2, hP = torch.autograd.grad(out, (x1, x2, P), g, allow_unused=True)
            torch.cuda.synchronize()
    finally:
        ops.PLANES.managed = was
    close(out, ref, rtol=1e-5, atol_scale=2e-6)
    close(hx1, gx1, rtol=1e-5, atol_scale=2e-6)
    close(hx2, gx2, rtol=1e-5, atol_scale=2e-6)
    got = (buf - 0.25).view(N, Kt)
    routed = [(lo, hi) for lo, hi in ((0, a), (a, Kt)) if (hi - lo) % 4 == 0 and (hi - lo) > 64]
    assert routed, "at least the wide range is one for the matrix-core kernels"
    for lo, hi in routed:
        close(got[:, lo:hi], gP.squeeze(-1)[:, lo:hi], rtol=2e-5, atol_scale=5e-6)
    if len(routed) == 2:
        assert hP is None and P.data_ptr() in scope.routed
    else:                                                                              # the narrow range went through autograd: the two add up
        assert hP is not None
        close(got + hP.squeeze(-1), gP.squeeze(-1), rtol=2e-5, atol_scale=5e-6)
    names = [e.key for e in prof.key_averages()]
    assert not any("reduce_kernel" in n_ for n_ in names), names                      # no torch reduction of weight-gradient partials
    if len(routed) == 2:
        assert not any("FillFunctor" in n_ for n_ in names), names                     # no zero-fill of the slice backward
