"""This repo's Point_MAE_unify against fixtures produced by the REFERENCE's own classes
(oracle/gen_golden.py).  CPU run: grouping served by the oracle through test injection."""
import numpy as np
import pytest
import torch

import _seeded
from models import build_model_from_cfg, MODELS, upp_layers
from utils.config import builtin_cfg

PEFT_KEYS = ['downstream_adapter', 'downstream_adapter1', 'downstream_prompts', 'bnorm', 'cls_pos', 'cls_token',
             'cls_head_finetune']


@pytest.fixture(scope="module")
def model():
    m = build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)
    return _seeded.fill(m).eval()


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_parameter_budget_and_schema(model):
    assert sum(p.numel() for p in model.parameters()) == 30_416_479
    sd = model.state_dict()
    assert len(sd) == 537
    trainable = sum(p.numel() for n, p in model.named_parameters() if any(k in n for k in PEFT_KEYS))
    assert trainable == 619_176
    assert sd['blocks.blocks.0.attn.qkv.weight'].shape == (1152, 384) and 'blocks.blocks.0.attn.qkv.bias' not in sd
    assert sd['encoder.first_conv.0.weight'].shape == (128, 3, 1)
    assert sd['blocks.blocks.5.downstream_prompts'].shape == (10, 384) and 'blocks.blocks.6.downstream_prompts' not in sd
    assert 'blocks.blocks.2.rectify_adapter.ln1.weight' in sd and 'blocks.blocks.3.rectify_adapter.ln1.weight' not in sd
    assert 'MAE_decoder.blocks.3.pretask_adapter.ln2.bias' in sd
    assert sd['rectify_prompter.propagation1.mlp_convs.0.weight'].shape == (32, 59, 1)
    assert MODELS.get('Point_MAE_unify') is type(model)


def test_logits_match_reference_clean_and_noisy(model, oracle_ops, golden):
    g = golden['upp_model']
    with torch.no_grad():
        lc = model(_seeded.unit_ball_clouds(2, 1024, 0))
        ln = model(_seeded.noisy_clouds(2, 1024, 0), completion_prompt=True, denoise=True, point_num=1024)
    np.testing.assert_allclose(lc.numpy(), g['logits_clean'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ln.numpy(), g['logits_noisy'], rtol=1e-5, atol=1e-5)


def test_peft_gradients_match_reference(model, oracle_ops, golden):
    g = golden['upp_model']
    for n, p in model.named_parameters():
        p.requires_grad_(any(k in n for k in PEFT_KEYS))
        p.grad = None
    logits = model(_seeded.noisy_clouds(2, 1024, 0), completion_prompt=True, denoise=True, point_num=1024)
    loss, _ = model.get_loss_acc(logits, T(g['labels']))
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-5)
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    assert sorted(grads) == list(g['grad_names'])
    norms = np.array([grads[n].norm().item() for n in g['grad_names']])
    np.testing.assert_allclose(norms, g['grad_norms'], rtol=2e-4, atol=1e-7)
    for k in g.files:
        if k.startswith('grad::'):
            np.testing.assert_allclose(grads[k[6:]].numpy(), g[k], rtol=1e-4, atol=1e-6)
    for p in model.parameters():
        p.requires_grad_(True)
        p.grad = None


STAGE2_KEYS = ['downstream_adapter', 'downstream_adapter1', 'downstream_prompts', 'dense_pred', 'mask_token', 'rectify_prompter',
               'shape_pred', 'coarse_pred', 'predict_token_generator', 'mask_prompter', 'mask_token_generator']   # runner_module.py:232-238


def test_stage2_joint_optimisation_gradients_match_reference(model, oracle_ops, golden):
    """Second half of the recipe (reference tools/runner_module.py:230-244): the prompter heads train, so the gradient has to
    travel back through grouping + patch embedding of the prompted cloud, the FPS gather, rebuild_points, the frozen decoder
    and backbone paths and the rectification offsets."""
    g = golden['upp_stage2']
    for n, p in model.named_parameters():
        p.requires_grad_(any(k in n for k in STAGE2_KEYS))
        p.grad = None
    logits = model(_seeded.noisy_clouds(2, 1024, 0), completion_prompt=True, denoise=True, point_num=1024)
    loss, _ = model.get_loss_acc(logits, T(g['labels']))
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), g['logits'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-5)
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    assert sorted(grads) == list(g['grad_names']) and len(grads) == 125
    norms = np.array([grads[n].norm().item() for n in g['grad_names']])
    np.testing.assert_allclose(norms, g['grad_norms'], rtol=5e-4, atol=1e-7)
    for k in g.files:
        if k.startswith('grad::'):
            ref = g[k]
            # (a long f32 chain through two prompters, three encoder passes and 31 block passes, in re-associated but
            # equivalent formulations: 1e-3 of the array's scale)
            np.testing.assert_allclose(grads[k[6:]].numpy(), ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max(), err_msg=k)
    for p in model.parameters():
        p.requires_grad_(True)
        p.grad = None


def _pick(grads, key):
    kind, name = key.split('::', 1)
    a = grads[name].detach().double().cpu().numpy()
    return a[::4, ::4] if kind == 'sampled4' else a


def test_stage2_function_is_the_reference_function_in_f64(oracle_ops, golden):
    """The stage-2 step with THIS repository's formulation evaluated in float64 against the reference's own classes evaluated in
    float64 (oracle/gen_golden.py stage2_f64; index sets from the f32 image of the coordinates on both sides): loss, all 125 gradient
    norms and the kept arrays agree to 1e-10 (measured 2e-15 ... 7e-15) -- the two formulations are the same function, term for term.
    What the f32 evaluations (CPU above, HIP in test_gpu_model.py) sit from the f32 fixture on the geometry-path gradients, 2.5e-4 ... 5e-4,
    is ONE max-pool arg-max flip in one group of the last patch embedding (4 of the 12,288 entries of the gradient w.r.t. its input are
    off by > 1e-3 of the scale, the median entry by 2e-8; the reference's own f32 run happens to have no flip: 1.5e-6 from its f64 run)."""
    g = golden['upp_stage2_f64']
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).eval()
    for n, p in m.named_parameters():
        p.requires_grad_(any(k in n for k in STAGE2_KEYS))
    m = m.double()
    logits = m(_seeded.noisy_clouds(2, 1024, 0).double(), completion_prompt=True, denoise=True, point_num=1024)
    loss, _ = m.get_loss_acc(logits, torch.tensor([3, 17]))
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), g['logits'], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(loss.item(), g['loss'], rtol=1e-12)
    grads = {n: p.grad for n, p in m.named_parameters() if p.requires_grad and p.grad is not None}
    assert sorted(grads) == list(g['grad_names'])
    norms = np.array([grads[n].norm().item() for n in g['grad_names']])
    np.testing.assert_allclose(norms, g['grad_norms'], rtol=1e-10, atol=1e-14)
    for k in g.files:
        if '::' in k:
            ref, got = g[k], _pick(grads, k)
            assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref), k


def test_group_outputs_and_index_layout(model, oracle_ops, golden):
    m = golden['upp_modules']
    nb, center, idx, cidx = model.group_divider(T(m['group_pts']), require_index=True, gather_idx=False)
    np.testing.assert_array_equal(idx.numpy(), m['group_idx'])              # flat, + b*N offsets, int64
    np.testing.assert_array_equal(cidx.numpy(), m['group_center_idx'])
    assert idx.dtype == torch.int64 and cidx.dtype == torch.int64
    np.testing.assert_array_equal(center.numpy(), m['group_center'])
    np.testing.assert_array_equal(nb.numpy(), m['group_neighborhood'])


def test_modules_match_reference(model, golden):
    m = golden['upp_modules']
    blk = model.blocks.blocks
    with torch.no_grad():
        np.testing.assert_allclose(model.encoder(T(m['group_neighborhood'])).numpy(), m['encoder_out'], rtol=1e-5, atol=2e-6)
        x = T(m['attn_in'])
        np.testing.assert_allclose(blk[0].attn(x).numpy(), m['attn_out'], rtol=1e-5, atol=1e-6)   # 1e-5 rel: north_star
        np.testing.assert_allclose(blk[0].mlp(x).numpy(), m['mlp_out'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(blk[0].downstream_adapter(x).numpy(), m['adapter_out'], rtol=1e-5, atol=1e-6)
        kw = dict(path='downstream', downstream_adapter=True, downstream_prompts=True, classification=True,
                  center1=T(m['group_center']), center1_idx=T(m['block_c1_idx']), center2=T(m['block_center2']),
                  center2_idx=T(m['block_c2_idx']), gather_idx=False, prompt_propagation_after=True)
        xb = T(m['block_in'])
        # block 0 inserts 10 prompts and propagates with the reference's stride-64-into-stride-74 indexing
        np.testing.assert_allclose(blk[0](xb, **kw).numpy(), m['block0_out'], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(blk[7](xb, **kw).numpy(), m['block7_out'], rtol=1e-5, atol=2e-6)
        xr = T(m['rect_in'])
        np.testing.assert_allclose(blk[1](xr, path='rectify', rectify_adapter=True, rectify_prompts=True, rectify_depth=3).numpy(),
                                   m['block1_rectify_out'], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(blk[4](xr, path='pretask', pretask_adapter=True, pretask_prompts=True, pretask_depth=6).numpy(),
                                   m['block4_pretask_out'], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(model.MAE_decoder(T(m['dec_in']), T(m['dec_pos']), 32, pretask_adapter=True, path='pretask').numpy(),
                                   m['dec_out'], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(upp_layers.propagate(T(m['group_center']), T(m['block_center2']), T(m['prop_p1']), T(m['prop_p2']),
                                                        de_neighbors=8, dist_e=1e-3).numpy(), m['prop_out'], rtol=1e-5, atol=2e-6)


def test_block_gather_idx_variant_and_rectify_prompter(model, oracle_ops, golden):
    m = golden['upp_modules']
    with torch.no_grad():
        lvl2 = upp_layers.Group(num_group=32, group_size=8)
        center = T(m['group_center'])
        _, c2, c1g, c2g = lvl2(center, require_index=True, gather_idx=True)
        assert c1g.shape == (2, 32, 8) and c2g.dtype == torch.int64
        kw = dict(path='downstream', downstream_adapter=True, downstream_prompts=True, classification=True,
                  center1=center, center1_idx=c1g, center2=c2, center2_idx=c2g, gather_idx=True, prompt_propagation_after=True)
        np.testing.assert_allclose(model.blocks.blocks[0](T(m['block_in']), **kw).numpy(), m['block0_gather_out'], rtol=1e-5, atol=2e-6)
        out = model.rectify_prompter(T(m['rp_pts']), T(m['rp_center']), T(m['rp_tokens']))
        np.testing.assert_allclose(out.numpy(), m['rp_out'], rtol=1e-5, atol=2e-6)


def test_checkpoint_key_rewrites(model, tmp_path):
    sd = {("module.MAE_encoder." + k if k.startswith("blocks.") else "base_model." + k): v for k, v in model.state_dict().items()}
    path = tmp_path / "ckpt.pth"
    torch.save({'base_model': sd}, path)
    fresh = build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)
    res = fresh.load_model_from_ckpt(str(path))
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(fresh.state_dict()['blocks.blocks.3.mlp.fc1.weight'], model.state_dict()['blocks.blocks.3.mlp.fc1.weight'])
