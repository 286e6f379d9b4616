"""Checkpoint wire format and the evaluation loops: files are interchangeable with the reference's
tools/builder.py layout; backbone checkpoints go through the models' key rewrites."""
import os

import numpy as np
import pytest
import torch

import _seeded
from models import build_model_from_cfg
from utils import checkpoint
from utils.config import builtin_cfg


@pytest.fixture(scope="module")
def model():
    return _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).eval()


def test_checkpoint_round_trip_and_reference_layouts(model, tmp_path):
    opt = torch.optim.AdamW([p for p in model.parameters()][:4], lr=1e-3)
    path = checkpoint.save_checkpoint(model, opt, 7, str(tmp_path / "exp" / "ckpt-last.pth"), best_metrics={'acc': 91.5})
    blob = torch.load(path, map_location='cpu')
    assert set(blob) == {'base_model', 'optimizer', 'epoch', 'best_metrics'} and blob['epoch'] == 7
    assert list(blob['base_model']) == list(model.state_dict())
    other = build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)
    assert checkpoint.resume_model(other, str(tmp_path / "exp")) == (8, {'acc': 91.5})
    for k, v in model.state_dict().items():
        assert torch.equal(other.state_dict()[k], v)
    assert checkpoint.resume_model(other, str(tmp_path / "nowhere")) == (0, 0)
    # DistributedDataParallel prefixes and the {'model': ...} layout of third-party backbones
    torch.save({'model': {'module.' + k: v for k, v in model.state_dict().items()}, 'epoch': 3}, tmp_path / "ddp.pth")
    third = build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)
    assert checkpoint.load_model(third, str(tmp_path / "ddp.pth")) == (3, 'No Metrics')
    assert torch.equal(third.state_dict()['cls_token'], model.state_dict()['cls_token'])
    with pytest.raises(NotImplementedError):
        checkpoint.load_model(third, str(tmp_path / "missing.pth"))
    torch.save({'weights': {}}, tmp_path / "bad.pth")
    with pytest.raises(RuntimeError, match="mismatch of ckpt weight"):
        checkpoint.load_model(third, str(tmp_path / "bad.pth"))


def test_backbone_checkpoint_key_rewrites(model, tmp_path):
    """models/Point_MAE_unify.py:505-536: 'module.', '_block', 'MAE_encoder.', 'base_model.' are stripped, strict=False."""
    sd = model.state_dict()
    src = {'module.MAE_encoder.encoder.first_conv.0.weight': sd['encoder.first_conv.0.weight'] + 1,
           'MAE_encoder.blocks.blocks.0.attn.qkv.weight': sd['blocks.blocks.0.attn.qkv.weight'] + 2,
           'base_model.norm.weight': sd['norm.weight'] + 3,
           'MAE_decoder_block.blocks.0.norm1.weight': sd['MAE_decoder.blocks.0.norm1.weight'] + 4,
           'something.else': torch.zeros(1)}
    torch.save({'base_model': src}, tmp_path / "backbone.pth")
    fresh = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model))
    res = fresh.load_model_from_ckpt(str(tmp_path / "backbone.pth"))
    tracked = sum(k.endswith('num_batches_tracked') for k in sd)          # BatchNorm does not report these as missing
    assert res.unexpected_keys == ['something.else'] and len(res.missing_keys) == len(sd) - 4 - tracked
    new = fresh.state_dict()
    assert torch.equal(new['encoder.first_conv.0.weight'], sd['encoder.first_conv.0.weight'] + 1)
    assert torch.equal(new['blocks.blocks.0.attn.qkv.weight'], sd['blocks.blocks.0.attn.qkv.weight'] + 2)
    assert torch.equal(new['norm.weight'], sd['norm.weight'] + 3)
    assert torch.equal(new['MAE_decoder.blocks.0.norm1.weight'], sd['MAE_decoder.blocks.0.norm1.weight'] + 4)


def test_validate_and_vote_on_the_oracle_ops(model, oracle_ops):
    from utils import evaluate
    torch.manual_seed(0)
    batches = [(_seeded.unit_ball_clouds(2, 1300, seed=s), torch.tensor([s, s + 1])) for s in (1, 2)]
    acc = evaluate.validate(model, batches, 1024)
    assert 0.0 <= float(acc) <= 100.0
    with torch.no_grad():
        want = torch.cat([model(oracle_ops["fps_gather"](p, 1024)[0]).argmax(-1) for p, _ in batches])
    labels = torch.cat([l for _, l in batches])
    np.testing.assert_allclose(float(acc), float((want == labels).sum()) / 4 * 100)
    g = torch.Generator().manual_seed(3)
    v1 = evaluate.test_vote(model, batches, 1024, times=2, transform=None, generator=g)
    g = torch.Generator().manual_seed(3)
    v2 = evaluate.test_vote(model, batches, 1024, times=2, transform=None, generator=g)
    assert float(v1) == float(v2) and 0.0 <= float(v1) <= 100.0
    with pytest.raises(NotImplementedError):
        evaluate.test_vote(model, batches, 2048)


@pytest.mark.gpu
def test_flat_adamw_state_dict_matches_torch_adamw_layout():
    from upp_hip.train import TrainStep, freeze_for_peft, make_adamw
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda()
    freeze_for_peft(m)
    ref_opt = make_adamw(m)
    ts = TrainStep(m, (2, 1096, 3), use_graph=False)
    ts.step(_seeded.noisy_clouds(2, 1024, 1).cuda(), torch.tensor([1, 2], device='cuda'))
    sd = ts.opt.state_dict()
    ref_sd = ref_opt.state_dict()
    assert [g['params'] for g in sd['param_groups']] == [g['params'] for g in ref_sd['param_groups']]
    assert [g['weight_decay'] for g in sd['param_groups']] == [0.0, 0.05]
    params = [p for g in ref_opt.param_groups for p in g['params']]
    for i, p in enumerate(params):
        assert sd['state'][i]['exp_avg'].shape == p.shape and float(sd['state'][i]['step']) == 1.0
    ref_opt.load_state_dict(sd)                       # torch accepts the layout
    m2 = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda()
    freeze_for_peft(m2)
    ts2 = TrainStep(m2, (2, 1096, 3), use_graph=False)
    ts2.opt.load_state_dict(sd)
    assert torch.equal(ts2.opt.m, ts.opt.m) and torch.equal(ts2.opt.v, ts.opt.v) and float(ts2.opt.state[0]) == 1.0


@pytest.mark.gpu
def test_vote_on_gpu(model):
    from utils import evaluate
    m = model.cuda()
    try:
        batches = [(_seeded.unit_ball_clouds(4, 1500, seed=s).cuda(), torch.tensor([s, 2, 3, 4], device='cuda')) for s in (1, 2)]
        acc = evaluate.test_vote(m, batches, 1024, times=3)
        assert 0.0 <= float(acc) <= 100.0
        assert 0.0 <= float(evaluate.validate(m, batches, 1024, noisy=True)) <= 100.0
    finally:
        model.cpu()
