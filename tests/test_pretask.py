"""Point_MAE_pretask_dev (prompter pre-training model + the three Chamfer-L1 terms of its recipe) against
the fixture produced by the reference's models/Point_MAE_pretask_dev.py with the same weights and inputs (train mode, all
dropout / drop-path probabilities 0; pytorch3d.knn_points served by the oracle kNN in the generator)."""
import os
import sys

import numpy as np
import pytest
import torch

import _seeded
from models import build_model_from_cfg, MODELS
from models.Point_MAE_pretask_dev import pretask_losses
from utils.config import builtin_cfg

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir, "oracle"))


def _inputs():
    gt = _seeded.unit_ball_clouds(2, 1280, seed=31)
    partial, cropping = gt[:, :1024].contiguous(), gt[:, 1024:].contiguous()
    noise = _seeded.noisy_clouds(2, 1024, seed=32)[:, 1024:1076].contiguous()
    return gt, partial, cropping, torch.cat([partial, noise], dim=1).contiguous()


def _deterministic_train(model):
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, 'drop_prob'):
            m.drop_prob = 0.0
    return model


@pytest.fixture(scope="module")
def pretask():
    return _seeded.fill(build_model_from_cfg(builtin_cfg('pretask').model))


def _run(model, golden, dev, rtol):
    g = golden['pretask']
    _seeded.fill(model)                               # train-mode BatchNorm moves the running statistics: start from the seed state
    _deterministic_train(model)
    gt, partial, cropping, points = (t.to(dev) for t in _inputs())
    for p in model.parameters():
        p.requires_grad_(True); p.grad = None
    total, terms = pretask_losses(model, gt, partial, cropping, points, point_num=1024)
    total.backward()
    for k_fix, k_term in (('coarse', 'cropping_coarse'), ('crop_dense', 'cropping_dense'), ('dense', 'dense'), ('noise_loss', 'noise')):
        np.testing.assert_allclose(terms[k_term].item(), g[k_fix], rtol=rtol)
    np.testing.assert_allclose(terms['recall'].item(), g['recall'], rtol=1e-6)
    np.testing.assert_allclose(total.item(), g['loss'], rtol=rtol)
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert sorted(grads) == list(g['grad_names'])
    norms = np.array([grads[n].norm().item() for n in g['grad_names']])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=50 * rtol, atol=3e-6)   # biases in front of a BatchNorm have ~0 gradient (rounding noise)
    model.eval()
    with torch.no_grad():
        center, rebuild = model(partial, point_num=1024, train_with_gaussian=False)
    np.testing.assert_allclose(center.cpu().numpy(), g['center_eval'], rtol=20 * rtol, atol=20 * rtol)
    np.testing.assert_allclose(rebuild[:, :128].cpu().numpy(), g['rebuild_eval_head'], rtol=20 * rtol, atol=20 * rtol)
    for p in model.parameters():
        p.grad = None


def test_schema(pretask, golden):
    g = golden['pretask']
    assert sum(p.numel() for p in pretask.parameters()) == int(g['n_params']) == 29_809_591
    assert len(pretask.state_dict()) == int(g['n_keys']) == 441
    assert MODELS.get('Point_MAE_pretask_dev') is type(pretask)
    sd = pretask.state_dict()
    assert sd['coarse_pred.2.weight'].shape == (96, 384) and 'rectify_prompter.score_head.3.bias' in sd
    assert not any(k.startswith('cls_') for k in sd)


def test_losses_and_gradients_match_reference(pretask, oracle_ops, golden):
    _run(pretask, golden, 'cpu', 2e-5)


@pytest.mark.gpu
def test_pretask_on_gpu_matches_fixture_and_trains(pretask, golden):
    m = pretask.cuda()
    try:
        # measured on MI355X (tools/micro/recipe_tolerance.py): loss terms within 1.2e-7, gradient norms (> 1e-4) within 2.8e-4 -> 2e-5 / 1e-3
        _run(m, golden, 'cuda', 2e-5)
        _seeded.fill(m).train()
        gt = _seeded.unit_ball_clouds(8, 1280, seed=3).cuda()
        partial, cropping = gt[:, :1024].contiguous(), gt[:, 1024:].contiguous()
        points = torch.cat([partial, _seeded.noisy_clouds(8, 1024, seed=4)[:, 1024:1076].cuda()], dim=1).contiguous()
        opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            total, _ = pretask_losses(m, gt, partial, cropping, points, point_num=1024)
            total.backward()
            opt.step()
            losses.append(total.item())
        assert all(np.isfinite(losses))
    finally:
        pretask.eval().cpu()
