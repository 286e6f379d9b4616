"""Point_MAE_unify_seg (BASELINE config 5) against the fixture produced by the reference's class."""
import numpy as np
import pytest
import torch

import _seeded
from models import build_model_from_cfg, MODELS
from utils.config import builtin_cfg


def _inputs():
    spts = _seeded.noisy_clouds(2, 1552, seed=11)
    lpts = _seeded.unit_ball_clouds(2, 2048, seed=12)
    return spts, lpts


@pytest.fixture(scope="module")
def seg():
    m = build_model_from_cfg(builtin_cfg('unify_shapenetpart_seg').model)
    return _seeded.fill(m).eval()


def _check(model, golden, dev, rtol, atol):
    g = golden['upp_seg']
    spts, lpts = _inputs()
    onehot = torch.from_numpy(g['onehot'])
    with torch.no_grad():
        logp = model(spts.to(dev), onehot.to(dev), label_points=lpts.to(dev), completion_prompt=True, denoise=True, point_num=1536)
        clean = model(lpts.to(dev), onehot.to(dev), label_points=None, completion_prompt=False, denoise=False, point_num=2048)
    assert logp.shape == (2, 2048, 50)
    np.testing.assert_allclose(logp[:, :256].cpu().numpy(), g['logp_head'], rtol=rtol, atol=atol)
    np.testing.assert_allclose(clean[:, :256].cpu().numpy(), g['logp_clean_head'], rtol=rtol, atol=atol)
    np.testing.assert_allclose(logp.double().sum(-1).cpu().numpy(), g['logp_sum'], rtol=rtol, atol=50 * atol)
    assert (logp.argmax(-1).cpu().numpy() == g['logp_argmax']).mean() > 0.999
    loss = model.get_loss(logp.reshape(-1, 50), torch.from_numpy(g['target']).reshape(-1).to(dev))
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=max(rtol, 1e-5))


def test_seg_schema(seg, golden):
    g = golden['upp_seg']
    assert sum(p.numel() for p in seg.parameters()) == int(g['n_params']) == 35_401_129
    assert len(seg.state_dict()) == int(g['n_keys'])
    sd = seg.state_dict()
    assert sd['seg_head.0.weight'].shape == (512, 3456, 1) and sd['propagation_0.mlp_convs.0.weight'].shape == (1536, 1155, 1)
    assert sd['label_conv.0.weight'].shape == (64, 16, 1) and sd['blocks.blocks.0.downstream_prompts'].shape == (1, 384)
    assert 'cls_token' not in sd and MODELS.get('Point_MAE_unify_seg') is type(seg)


def test_seg_log_probabilities_match_reference(seg, oracle_ops, golden):
    _check(seg, golden, 'cpu', 1e-5, 2e-5)


@pytest.mark.gpu
def test_seg_on_gpu_matches_reference_fixture(seg, golden):
    m = seg.cuda()
    try:
        _check(m, golden, 'cuda', 1e-5, 2e-5)
    finally:
        seg.cpu()


SEG_PEFT = ['downstream_adapter', 'downstream_prompts', 'label_conv', 'propagation_0', 'seg_head', 'propagation_1']   # reference tools/runner_unify_seg.py:143-146


def _deterministic_train(model):
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, 'drop_prob'):
            m.drop_prob = 0.0
    return model


def _train_step_check(model, golden, dev, rtol, arr_tol):
    """One training step of the part-segmentation recipe against the reference class's own step (oracle/gen_golden.py seg_train: TRAIN
    mode, batch-statistics BatchNorm, dropout 0): log-probabilities, loss, the gradient norm of all 104 trainable tensors, 13 gradient
    arrays element by element and every 16th row / column of the four large head weights."""
    g = golden['upp_seg_train']
    _seeded.fill(model)                               # train-mode BatchNorm moves the running statistics: start from the seed state
    _deterministic_train(model)
    for n, p in model.named_parameters():
        p.requires_grad_(any(k in n for k in SEG_PEFT)); p.grad = None
    spts, lpts = _inputs()
    onehot = torch.from_numpy(golden['upp_seg']['onehot']).to(dev)
    tgt = torch.from_numpy(golden['upp_seg']['target']).reshape(-1).to(dev)
    try:
        logp = model(spts.to(dev), onehot, label_points=lpts.to(dev), completion_prompt=True, denoise=True, point_num=1536)
        np.testing.assert_allclose(logp[:, :256].detach().cpu().numpy(), g['logp_head'], rtol=rtol, atol=2 * rtol)
        loss = model.get_loss(logp.reshape(-1, 50), tgt)
        loss.backward()
        np.testing.assert_allclose(loss.item(), g['loss'], rtol=max(rtol, 1e-5))
        grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}
        assert sorted(grads) == list(g['grad_names'])
        # Measured between two f32 formulations of the SAME step on the CPU (ours on torch-CPU against the reference's classes): the head,
        # the propagation and the adapters agree to <= 4e-4 of a gradient's norm; label_conv sits behind a BatchNorm over B = 2 rows (its
        # output is +-1 whatever the input: the input gradient is rounding noise amplified by 1/sigma) and differs by 0.5 %.
        names = list(g['grad_names'])
        norms = np.array([grads[n].norm().item() for n in names])
        loose = np.array(['label_conv' in n for n in names])
        np.testing.assert_allclose(norms[~loose], g['grad_norms'][~loose], rtol=3e-3, atol=3e-6)   # (biases in front of a BatchNorm: ~0)
        np.testing.assert_allclose(norms[loose], g['grad_norms'][loose], rtol=3e-2, atol=3e-6)
        # Element by element the two formulations differ through ReLU gates that flip on 1e-7 forward differences (measured on the head
        # alone with random inputs: 1e-3 relative L2 between the concat form and the split form of the SAME layer), so arrays are held to
        # a relative L2 distance: measured <= 8.2e-3 over three states of the kernels (label_conv 1.3e-2); a wrong formula is O(1).  A bias in front of a BatchNorm has an
        # analytically zero gradient: noise on both sides, bounded instead of compared.
        for k in g.files:
            if k.startswith('grad::') or k.startswith('sampled::'):
                ref = g[k]
                name = k.split('::', 1)[1]
                got = grads[name].cpu().numpy() if k.startswith('grad::') else grads[name].squeeze(-1)[::16, ::16].cpu().numpy()
                if name == 'propagation_0.mlp_convs.1.bias':
                    assert np.abs(got).max() < 1e-3 * np.abs(grads['propagation_0.mlp_bns.1.bias'].cpu().numpy()).max(), k
                    continue
                dist = np.linalg.norm((got - ref).ravel()) / np.linalg.norm(ref.ravel())
                assert dist < (4e-2 if 'label_conv' in name else arr_tol), (k, dist)
                # f64 arbitration (the reference's classes in float64, upp_seg_train_f64.npz): this f32 evaluation is no further from the
                # exact gradient than ten times the reference's own f32 evaluation is.  Measured over three states of the kernels
                # (tools/micro/seg_flip_noise.py): 0.15 ... 4.5 x where that is above 1e-4 -- each f32 evaluation is ONE draw of the ReLU-gate
                # flips under a 2 x 2048-row train-mode BatchNorm (dense: the median element moves with the L2 figure), and re-associating one
                # sum upstream (the patch embedding's per-group product changed kernels in round 3) redraws them; the arrays next to the loss
                # sit at 1e-5 ... 1e-4 on both sides.  An order-of-magnitude guard; the float64 test above is what pins the function.
                exact = golden['upp_seg_train_f64'][k]
                mine = np.linalg.norm((got - exact).ravel()) / np.linalg.norm(exact.ravel())
                theirs = np.linalg.norm((ref - exact).ravel()) / np.linalg.norm(exact.ravel())
                assert mine <= 10.0 * theirs + 1e-4, (k, mine, theirs)
    finally:
        for p in model.parameters():
            p.requires_grad_(True); p.grad = None
        _seeded.fill(model).eval()


def test_seg_train_step_matches_the_reference_step(seg, oracle_ops, golden):
    _train_step_check(seg, golden, 'cpu', 1e-5, 1.5e-2)


def test_seg_train_step_is_the_reference_function_in_f64(oracle_ops, golden):
    """The same training step, this repository's formulation in float64 against the reference's classes in float64
    (oracle/gen_golden.py seg_train_f64): loss, the 104 gradient norms and the kept / sampled arrays agree to 1e-9 (measured 2e-14 ...
    4e-13): the commuted first propagation layer, the split head layer, the BatchNorm-over-rows formulation are the reference's
    function.  The 1e-3 ... 3e-3 the f32 evaluations differ by is the step's conditioning (train-mode BatchNorm over 2 x 2048 rows
    behind ReLU gates): the reference's own f32 run sits as far from its f64 run (arbitrated in _train_step_check)."""
    g, g0 = golden['upp_seg_train_f64'], golden['upp_seg']
    m = _deterministic_train(_seeded.fill(build_model_from_cfg(builtin_cfg('unify_shapenetpart_seg').model)))
    for n, p in m.named_parameters():
        p.requires_grad_(any(k in n for k in SEG_PEFT))
    m = m.double()
    spts, lpts = _inputs()
    logp = m(spts.double(), torch.from_numpy(g0['onehot']).double(), label_points=lpts.double(), completion_prompt=True, denoise=True, point_num=1536)
    loss = m.get_loss(logp.reshape(-1, 50), torch.from_numpy(g0['target']).reshape(-1))
    loss.backward()
    np.testing.assert_allclose(logp[:, :256].detach().numpy(), g['logp_head'], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-12)
    grads = {n: p.grad for n, p in m.named_parameters() if p.requires_grad and p.grad is not None}
    assert sorted(grads) == list(g['grad_names'])
    norms = np.array([grads[n].norm().item() for n in g['grad_names']])
    np.testing.assert_allclose(norms, g['grad_norms'], rtol=1e-9, atol=1e-13)      # (biases in front of a BatchNorm: 1e-15 on both sides)
    for k in g.files:
        if '::' in k and 'propagation_0.mlp_convs.1.bias' not in k:
            ref = g[k]
            name = k.split('::', 1)[1]
            got = grads[name].numpy() if k.startswith('grad::') else grads[name].squeeze(-1)[::16, ::16].numpy()
            assert np.linalg.norm((got - ref).ravel()) <= 1e-9 * np.linalg.norm(ref.ravel()), k


@pytest.mark.gpu
def test_seg_train_step_on_gpu_matches_the_reference_step(seg, golden):
    m = seg.cuda()
    try:
        _train_step_check(m, golden, 'cuda', 2e-5, 1.5e-2)
    finally:
        seg.cpu()


@pytest.mark.gpu
def test_seg_head_layers_at_the_benched_row_count_use_the_tall_kernel_and_agree_with_the_short_one():
    """At B = 32 the head's point rows (65,536) cross the register-tiled kernel's threshold, at the fixture's B = 2 (4,096) they do not:
    the two kernels share one summation order (ks = 1), so the same rows give the same bits either way -- forward, data gradient, and
    the weight gradient summed over its runs to 1e-5."""
    from upp_hip import functional as HF, ops, _abi
    lib = _abi.load()
    g = torch.Generator(device='cuda').manual_seed(3)
    for N, K in ((1024, 1536), (512, 1024), (256, 512)):
        x = torch.randn(65536, K, device='cuda', generator=g)
        w = (torch.randn(N, K, device='cuda', generator=g) * K ** -0.5).requires_grad_(True)
        b = torch.randn(N, device='cuda', generator=g)
        assert lib.upp_linear_tile(65536, N, K) & 0x10000 and not lib.upp_linear_tile(4096, N, K) & 0x10000
        tall = HF.linear(x, w, b)
        short_tile = lib.upp_linear_tile(4096, N, K)
        if (short_tile >> 4) & 15 == 1:               # the short shape's choice also keeps the contraction whole: bit for bit
            assert torch.equal(tall[:4096], ops.linear_f32(x[:4096], w.detach(), b, ops.LIN_BIAS))
        ref = torch.nn.functional.linear(x[:8192], w.detach(), b)
        np.testing.assert_allclose(tall[:8192].detach().cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=2e-6 * ref.abs().max().item())
        gy = torch.randn(65536, N, device='cuda', generator=g)
        xg = x.clone().requires_grad_(True)
        HF.linear(xg, w, b).backward(gy)
        ref_dx = gy[:8192] @ w.detach()
        np.testing.assert_allclose(xg.grad[:8192].cpu().numpy(), ref_dx.cpu().numpy(), rtol=1e-5, atol=2e-6 * ref_dx.abs().max().item())
        ref_dw = (gy.double().t() @ x.double())
        np.testing.assert_allclose(w.grad.double().cpu().numpy(), ref_dw.cpu().numpy(), rtol=1e-5, atol=2e-6 * ref_dw.abs().max().item())


@pytest.mark.gpu
def test_seg_train_step_runs_on_gpu(seg):
    from upp_hip.train import freeze_for_peft
    m = seg.cuda().train()
    try:
        freeze_for_peft(m, ['downstream_adapter', 'downstream_prompts', 'bnorm', 'label_conv', 'propagation_0', 'seg_head'])
        spts, lpts = _inputs()
        onehot = torch.zeros(2, 16, device='cuda'); onehot[:, 2] = 1
        logp = m(spts.cuda(), onehot, label_points=lpts.cuda(), completion_prompt=True, denoise=True, point_num=1536)
        loss = m.get_loss(logp.reshape(-1, 50), torch.randint(0, 50, (2 * 2048,), device='cuda'))
        loss.backward()
        assert torch.isfinite(loss)
        assert m.seg_head[0].weight.grad is not None and torch.isfinite(m.seg_head[0].weight.grad).all()
        assert m.blocks.blocks[0].downstream_prompts.grad.abs().sum() > 0
    finally:
        for p in m.parameters():
            p.requires_grad_(True); p.grad = None
        seg.eval().cpu()


@pytest.mark.gpu
def test_seg_train_step_on_gpu_with_the_hip_runs_own_discrete_choices(oracle_ops, golden):
    """The flip-free comparison of the segmentation training step (round-4 verdict 6d; the stage-2 analogue is
    tests/test_gpu_model.py::test_stage2_gradients_with_the_hip_runs_own_discrete_choices).  Two f32 evaluations of this step differ by
    1e-3 ... 1e-2 because train-mode BatchNorm over 2 x 2048 rows puts thousands of pre-activations within 1e-7 of a ReLU gate and every
    evaluation redraws those gates -- which says nothing about the kernels.  Here an instrumented HIP run records EVERY discrete choice of
    its forward (upp_layers.POOL_TRACE: FPS picks, neighbour lists, interpolation lists, max-pool arg-maxes, the rectify prompter's
    ranking, and -- new -- the mask of every ReLU / LeakyReLU) and CPU evaluations of the torch formulation REPLAY them in float32 and
    float64.  With the gates pinned the step is a smooth function, and the HIP path must agree with the f32 torch formulation to 2e-5 of
    every gradient array's scale + 2 x that formulation's own distance from float64 (measured: 2.3e-5 on an adapter weight where the torch
    f32 evaluation itself sits 1.6e-5 from float64; label_conv, behind a BatchNorm over B = 2 rows, 1.9e-4 / 1.1e-4 -- against 1e-3 ...
    3e-2 between two f32 evaluations with free gates)."""
    from models import upp_layers as L
    from upp_hip import functional as HF
    spts, lpts = _inputs()
    onehot = torch.from_numpy(golden['upp_seg']['onehot'])
    tgt = torch.from_numpy(golden['upp_seg']['target']).reshape(-1)

    def fresh(dtype, dev):
        m = _deterministic_train(_seeded.fill(build_model_from_cfg(builtin_cfg('unify_shapenetpart_seg').model)))
        for n, p in m.named_parameters():
            p.requires_grad_(any(k in n for k in SEG_PEFT))
        return m.to(dtype).to(dev)

    def grads(m, dtype, dev):
        logp = m(spts.to(dtype).to(dev), onehot.to(dtype).to(dev), label_points=lpts.to(dtype).to(dev), completion_prompt=True, denoise=True, point_num=1536)
        loss = m.get_loss(logp.reshape(-1, 50), tgt.to(dev))
        loss.backward()
        return {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.requires_grad and p.grad is not None}, loss.item()

    saved = dict(L.OPS)
    L.OPS.update(fps_gather=HF.fps_gather, knn_group=HF.knn_group)        # (the HIP grouping primitives, whatever a fixture put into the table)
    trace = {'mode': 'record', 'items': []}
    try:
        L.POOL_TRACE = trace
        traced, loss_t = grads(fresh(torch.float32, 'cuda'), torch.float32, 'cuda')
    finally:
        L.POOL_TRACE = None
        L.OPS.clear(); L.OPS.update(saved)
    sites = {k[0] for k, _ in trace['items']}
    assert {'group.fps', 'group.knn', 'bn_rows.relu', 'label_conv', 'seg.global_max', 'rectify.order', 'misc.fps'} <= sites, sites

    def cpu(dtype):
        L.POOL_TRACE = {'mode': 'replay', 'items': trace['items']}
        try:
            out = grads(fresh(dtype, 'cpu'), dtype, 'cpu')
            assert L.POOL_TRACE['pos'] == len(trace['items'])               # same sites, same order
        finally:
            L.POOL_TRACE = None
        return out
    (c32, loss32), (c64, loss64) = cpu(torch.float32), cpu(torch.float64)
    assert sorted(c64) == sorted(traced) == sorted(c32)
    np.testing.assert_allclose([loss_t, loss32], loss64, rtol=5e-6)

    def rel(a, b):
        return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    worst, window, bad = 0.0, 0.0, []
    for n in c32:
        if n == 'propagation_0.mlp_convs.1.bias' or (n.endswith('.bias') and c64[n].abs().max().item() < 1e-9):
            continue                                              # (a bias in front of a BatchNorm: analytically zero, rounding noise on every side)
        gap32, err = rel(c32[n], c64[n]), rel(traced[n], c32[n])
        worst, window = max(worst, err), max(window, gap32)
        bad += [(n, err, gap32)] if err > 2e-5 + 2.0 * gap32 else []          # (triangle inequality: two f32 evaluations, each gap32 from float64)
    assert not bad, bad
    print("seg training step with replayed choices: hip vs cpu f32 worst %.2e; cpu f32 vs f64 worst %.2e over %d sites" % (worst, window, len(trace['items'])))
