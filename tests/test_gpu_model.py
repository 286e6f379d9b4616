"""End-to-end on the GPU: Point_MAE_unify with the HIP grouping kernels against the fixtures
produced by the reference's own classes (FPS / kNN indices must agree bit for bit for the
logits to agree)."""
import numpy as np
import pytest
import torch

import _seeded
from models import build_model_from_cfg
from utils.config import builtin_cfg

pytestmark = pytest.mark.gpu

PEFT_KEYS = ['downstream_adapter', 'downstream_adapter1', 'downstream_prompts', 'bnorm', 'cls_pos', 'cls_token',
             'cls_head_finetune']


@pytest.fixture(scope="module")
def model():
    m = build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)
    return _seeded.fill(m).eval().cuda()


def test_logits_match_reference_fixture(model, golden):
    g = golden['upp_model']
    with torch.no_grad():
        lc = model(_seeded.unit_ball_clouds(2, 1024, 0).cuda())
        ln = model(_seeded.noisy_clouds(2, 1024, 0).cuda(), completion_prompt=True, denoise=True, point_num=1024)
    # north_star's bar, 1e-5 on O(1) logits (f32 HIP kernels here, MKL on the CPU in the fixture; measured: 7e-7 / 1e-6 max abs,
    # tools/micro/logit_tolerance.py)
    np.testing.assert_allclose(lc.cpu().numpy(), g['logits_clean'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ln.cpu().numpy(), g['logits_noisy'], rtol=1e-5, atol=1e-5)


def test_modules_match_reference_fixture(model, golden):
    m = golden['upp_modules']
    T = lambda k: torch.from_numpy(m[k]).cuda()
    with torch.no_grad():
        nb, center, idx, cidx = model.group_divider(T('group_pts'), require_index=True, gather_idx=False)
        np.testing.assert_array_equal(idx.cpu().numpy(), m['group_idx'])
        np.testing.assert_array_equal(cidx.cpu().numpy(), m['group_center_idx'])
        np.testing.assert_array_equal(nb.cpu().numpy(), m['group_neighborhood'])
        np.testing.assert_allclose(model.encoder(nb).cpu().numpy(), m['encoder_out'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(model.blocks.blocks[0].attn(T('attn_in')).cpu().numpy(), m['attn_out'], rtol=1e-5, atol=1e-5)


def test_train_step_gradients_match_fixture(model, golden):
    g = golden['upp_model']
    for n, p in model.named_parameters():
        p.requires_grad_(any(k in n for k in PEFT_KEYS))
        p.grad = None
    logits = model(_seeded.noisy_clouds(2, 1024, 0).cuda(), completion_prompt=True, denoise=True, point_num=1024)
    loss, _ = model.get_loss_acc(logits, torch.from_numpy(g['labels']).cuda())
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-5)
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    assert sorted(grads) == list(g['grad_names'])
    norms = np.array([grads[n].norm().item() for n in g['grad_names']])
    np.testing.assert_allclose(norms, g['grad_norms'], rtol=2e-5, atol=1e-7)     # (measured 1e-6, tools/micro/recipe_tolerance.py)
    for k in g.files:                       # every full gradient array the fixture holds, element by element, from the HIP backward
        if k.startswith('grad::'):
            ref = g[k]
            # measured: 2.9e-6 of the array's scale, 1.8e-6 relative L2, 6.3e-5 relative on entries above 1 % of the scale
            np.testing.assert_allclose(grads[k[6:]].cpu().numpy(), ref, rtol=2e-4, atol=1e-5 * np.abs(ref).max(), err_msg=k)
    for p in model.parameters():
        p.requires_grad_(True)
        p.grad = None


STAGE2_KEYS = ['downstream_adapter', 'downstream_adapter1', 'downstream_prompts', 'dense_pred', 'mask_token', 'rectify_prompter',
               'shape_pred', 'coarse_pred', 'predict_token_generator', 'mask_prompter', 'mask_token_generator']   # runner_module.py:232-238


def _stage2_check(grads, loss, g):
    """Measured (tools/micro/recipe_tolerance.py): loss 6e-8, norms <= 3.8e-4, arrays <= 5.4e-4 of their scale / 4.5e-4 relative L2 on the
    gradients that come through the prompted geometry, 2e-6 on the others.  The geometry-path figure is ONE max-pool arg-max flip in one
    group of the last patch embedding between two f32 evaluations (tests/test_model_golden.py: in float64 this formulation and the
    reference's agree to 1e-14 on every array, and the CPU f32 path of this repository reproduces the HIP figures to three digits)."""
    np.testing.assert_allclose(loss, g['loss'], rtol=1e-6)
    assert sorted(grads) == list(g['grad_names'])
    norms = np.array([grads[n].norm().item() for n in g['grad_names']])
    np.testing.assert_allclose(norms, g['grad_norms'], rtol=1e-3, atol=1e-6)
    for k in g.files:
        if k.startswith('grad::'):
            ref = g[k]
            got = grads[k[6:]].cpu().numpy()
            assert np.linalg.norm(got - ref) <= 1e-3 * np.linalg.norm(ref), k
            np.testing.assert_allclose(got, ref, rtol=0, atol=1.5e-3 * np.abs(ref).max(), err_msg=k)


def test_stage2_joint_optimisation_gradients_match_fixture(model, golden):
    """The second stage of the recipe (reference tools/runner_module.py:230-244) on the GPU: prompter heads trainable, the
    gradient runs back through the prompting front-end (FPS gather, grouping, patch embedding, frozen decoder / backbone)."""
    g = golden['upp_stage2']
    for n, p in model.named_parameters():
        p.requires_grad_(any(k in n for k in STAGE2_KEYS))
        p.grad = None
    logits = model(_seeded.noisy_clouds(2, 1024, 0).cuda(), completion_prompt=True, denoise=True, point_num=1024)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g['logits'], rtol=1e-5, atol=1e-5)
    loss, _ = model.get_loss_acc(logits, torch.from_numpy(g['labels']).cuda())
    loss.backward()
    _stage2_check({n: p.grad for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}, loss.item(), g)
    for p in model.parameters():
        p.requires_grad_(True)
        p.grad = None


def test_stage2_gradients_with_the_hip_runs_own_discrete_choices(model, oracle_ops, golden):
    """Round-3 verdict, item 6: a stage-2 comparison without a flip allowance.  The forward makes discrete choices (FPS picks, neighbour
    lists, max-pool arg-maxes, the rectify prompter's ranking); upp_layers.trace_idx / max_over record them on a HIP run and REPLAY them in
    CPU evaluations of this repository's torch formulation (== the reference's classes to 1e-14 in float64: test_model_golden.py):
      * with the exact-f32 patch embedding (option EMBED_SPLIT_BF16 = 0) the product path (fused kernels, no instrument) and the instrumented
        HIP run both agree with the CPU *float32* evaluation to 2e-5 of every gradient array's scale (measured 5.6e-6 / 5.4e-6): the
        denoised coordinates then come out bit-identical on both sides;
      * against *float64* that HIP path is exactly as far as the CPU float32 evaluation is (8e-4 on the rectify prompter's BatchNorm
        parameters, both): the window is the f32 conditioning of the reference's own formula -- square_distance = |a|^2 + |b|^2 - 2ab
        cancels to +-1e-7 where a query coincides with a centre and the interpolation weight is 1 / (d + 1e-4) -- and no arg-max flip
        (with every choice replayed, gated and ungated float64 runs are identical);
      * the default product path (patch embedding on the bf16 pipe: features differ by 1e-7, the denoised coordinates by an ulp) moves
        exactly those ill-conditioned entries by the same 8e-4 and nothing else: per array it is asserted no further from the CPU float32
        evaluation than 2e-5 + 1.5 x that evaluation's own distance from float64."""
    from models import upp_layers as L
    from upp_hip import functional as HF
    import os
    if os.environ.get('UPP_SPLIT_BF16') == '0':
        pytest.skip("written for the shipped configuration: the un-instrumented runs make their own discrete choices, and on the exact-f32 "
                    "Linear kernels one of them flips on this input (the 1e-3 window of _stage2_check)")
    g = golden['upp_stage2_f64']
    pts, labels = _seeded.noisy_clouds(2, 1024, 0), torch.tensor([3, 17])
    saved = dict(L.OPS)

    def grads(m, x, y):
        for n, p in m.named_parameters():
            p.requires_grad_(any(k in n for k in STAGE2_KEYS))
            p.grad = None
        logits = m(x, completion_prompt=True, denoise=True, point_num=1024)
        loss, _ = m.get_loss_acc(logits, y)
        loss.backward()
        out = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.requires_grad and p.grad is not None}
        for p in m.parameters():
            p.requires_grad_(True)
            p.grad = None
        return out, loss.item()

    def worst(a, b):
        return max((a[n] - b[n]).abs().max().item() / max(b[n].abs().max().item(), 1e-30) for n in b)

    # (model is on the GPU: its grouping primitives are the HIP ones whatever the oracle_ops fixture put into the table)
    L.OPS.update(fps_gather=HF.fps_gather, knn_group=HF.knn_group)
    trace = {'mode': 'record', 'items': []}
    from upp_hip import ops
    try:
        default, loss_d = grads(model, pts.cuda(), labels.cuda())
        with ops.option("EMBED_SPLIT_BF16", 0):
            product, loss_p = grads(model, pts.cuda(), labels.cuda())
            L.POOL_TRACE = trace
            traced, loss_t = grads(model, pts.cuda(), labels.cuda())
    finally:
        L.POOL_TRACE = None
        L.OPS.clear(); L.OPS.update(saved)
    sites = {k[0] for k, _ in trace['items']}
    assert {'group.fps', 'group.knn', 'interp.knn', 'rectify.order', 'misc.fps', 'encoder.pool1', 'block.pooling', 'cls.max'} <= sites

    def cpu(dtype):
        m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).eval().to(dtype)
        L.POOL_TRACE = {'mode': 'replay', 'items': trace['items']}
        try:
            out = grads(m, pts.to(dtype), labels)
            assert L.POOL_TRACE['pos'] == len(trace['items'])                   # same sites, same order
        finally:
            L.POOL_TRACE = None
        return out
    (c32, loss32), (c64, loss64) = cpu(torch.float32), cpu(torch.float64)
    assert sorted(c64) == sorted(product) == sorted(traced) == list(g['grad_names'])
    np.testing.assert_allclose([loss_d, loss_p, loss_t, loss32], loss64, rtol=3e-6)
    e_prod, e_traced, e_prod64, e_c32 = worst(product, c32), worst(traced, c32), worst(product, c64), worst(c32, c64)
    print("stage-2 gradients, worst entry / array scale: product vs cpu f32 %.2e, traced hip vs cpu f32 %.2e, product vs f64 %.2e, "
          "cpu f32 vs f64 %.2e" % (e_prod, e_traced, e_prod64, e_c32))
    # Per array: within 2e-5 of the CPU float32 evaluation where that evaluation is within 2e-5 of float64 (the well-conditioned arrays:
    # their f32-f64 gap is ~1e-6); for the others -- everything behind the interpolation weights: the rectify prompter, mask_token -- within
    # 2e-5 + 1.5 x the window e_c32 that the CPU float32 evaluation itself shows against float64 (every f32 evaluation is one draw in it).  (With
    # the exact-f32 patch embedding and the split-bf16 Linear layers the denoised coordinates come out bit-identical on both sides and
    # the first two runs are within 5.6e-6 overall -- measured, printed above, not asserted: an ulp anywhere upstream ends it.)
    for name, run in (('exact-f32 patch embedding', product), ('instrumented', traced), ('default', default)):
        for n in c32:
            scale = max(c32[n].abs().max().item(), 1e-30)
            gap32 = (c32[n] - c64[n]).abs().max().item() / scale
            diff = (run[n] - c32[n]).abs() / scale
            err = diff.max().item()
            if name == 'default':
                # the default run's patch embedding (split-bf16 products, fused max-pool epilogue) makes its OWN arg-max choices -- they
                # cannot be replayed from the trace -- so it is one more f32 draw: an ulp upstream may flip one max-pool arg-max and move
                # the gradients behind it by ~6e-4 of their scale (seen on mask_token in round 6 when the fitted tile model changed the
                # summation split of an upstream GEMM; 7e-6 in rounds 4-5 -- mask_token is a (1,1,384) sum over every masked position, so
                # one flipped path shifts all of its entries; which array moves depends on the state earlier tests left in the module's
                # model: 5.9e-4 on mask_token in the full suite; 1.2e-3 on dense_pred.0.weight and 1.04e-3 relative L2 on the rectify
                # prompter's first point-wise layer when this file runs alone).  Asserted as a sanity window around ONE flip -- 3e-3 relative
                # L2 per array, no entry beyond 1e-2 of the array's scale (a wrong kernel shows at 1e-1) -- the flip-free statement is the
                # two replayed runs above.
                rel_l2 = ((run[n] - c32[n]).norm() / c32[n].norm().clamp_min(1e-30)).item()
                assert rel_l2 <= 3e-3 and err <= 1e-2, (name, n, rel_l2, err, gap32, e_c32)
                continue
            assert err <= 2e-5 + (1.5 * e_c32 if gap32 > 2e-5 else 0.0), (name, n, err, gap32, e_c32)
    assert e_prod64 <= 1.5 * e_c32 + 2e-5


def test_stage2_through_the_step_driver(golden):
    """TrainStep (one stream, lr = 0) on the stage-2 parameter list reproduces the fixture's gradients in its flat buffer."""
    from upp_hip.train import TrainStep, freeze_for_peft
    g = golden['upp_stage2']
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).eval().cuda()
    freeze_for_peft(m, STAGE2_KEYS)
    ts = TrainStep(m, (2, 1096, 3), lr=0.0, grad_clip=None)
    ts.step(_seeded.noisy_clouds(2, 1024, 0).cuda(), torch.from_numpy(g['labels']).cuda())
    torch.cuda.synchronize()
    names = {id(p): n for n, p in m.named_parameters()}
    grads = {names[id(p)]: v.clone() for p, v in zip(ts.trainable, ts.flat.views)}
    _stage2_check(grads, float(ts.loss), g)


def test_weights_loaded_into_a_captured_stage2_step_reach_its_cached_copies(golden):
    """load_state_dict into an already captured stage-2 step (the gradient runs through the patch embedding: the padded first-conv weight
    and its W^T copy are read by captured launches) + functional.refresh_caches(model): the next replay equals an eager step of a
    freshly built model with the new weights."""
    from upp_hip import functional as HF
    from upp_hip.train import TrainStep, freeze_for_peft
    g = golden['upp_stage2']
    x, y = _seeded.noisy_clouds(2, 1024, 0).cuda(), torch.from_numpy(g['labels']).cuda()
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).eval().cuda()
    freeze_for_peft(m, STAGE2_KEYS)
    ts = TrainStep(m, (2, 1096, 3), lr=0.0, grad_clip=None)
    ts.step(x, y)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(ts.loss), g['loss'], rtol=1e-4)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    for k in sd:
        if k.startswith('encoder.first_conv.0.') or k.startswith('encoder.second_conv.3.') or 'blocks.blocks.3.attn.qkv.weight' in k:
            sd[k] = sd[k] * 1.25
    m.load_state_dict(sd)
    HF.refresh_caches(m)
    ts.step(x, y)
    torch.cuda.synchronize()
    fresh = build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model).eval().cuda()
    fresh.load_state_dict(sd)
    freeze_for_peft(fresh, STAGE2_KEYS)
    loss, _ = fresh.get_loss_acc(fresh(x, completion_prompt=True, denoise=True, point_num=1024), y)
    assert abs(float(ts.loss) - float(g['loss'])) > 1e-4 * abs(float(g['loss']))          # the new weights matter
    np.testing.assert_allclose(float(ts.loss), loss.item(), rtol=2e-5)
    loss.backward()
    names = {id(p): n for n, p in m.named_parameters()}
    ref = dict(fresh.named_parameters())
    for p, v in zip(ts.trainable, ts.flat.views):
        r = ref[names[id(p)]].grad
        if r is not None and r.abs().max() > 0:
            np.testing.assert_allclose(v.cpu().numpy(), r.cpu().numpy(), rtol=2e-3, atol=2e-3 * r.abs().max().item(), err_msg=names[id(p)])


def test_auxiliary_reconstruction_losses_on_the_completion_prompters_points(model):
    """bench.py --workload cls_aux: Chamfer-L1 + EMD on the rebuilt points of the completion prompter, forward and
    backward (gradient w.r.t. rebuild_points), next to the classification forward."""
    from extensions.chamfer_dist import ChamferDistanceL1
    from emd import emd
    import oracle as O
    B = 2
    with torch.no_grad():
        model(_seeded.noisy_clouds(B, 1024, 0).cuda(), completion_prompt=True, denoise=True, point_num=1024)
    rb = model.aux['rebuild_points']
    assert tuple(rb.shape) == (B, 1024, 3)
    gt = _seeded.unit_ball_clouds(B, 1024, 0).cuda()
    x = rb.detach().clone().requires_grad_(True)
    cd = ChamferDistanceL1()(x, gt)
    em = emd()(x, gt)
    (cd + em).backward()
    assert torch.isfinite(x.grad).all() and x.grad.abs().sum().item() > 0
    d1, d2, _, _ = O.chamfer_fwd(rb.cpu().numpy(), gt.cpu().numpy())
    np.testing.assert_allclose(cd.item(), (np.sqrt(d1).mean() + np.sqrt(d2).mean()) / 2, rtol=1e-5)
