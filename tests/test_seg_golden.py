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
