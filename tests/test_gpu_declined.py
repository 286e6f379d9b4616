"""Nothing of the headline and part-segmentation steps falls back to a torch / library formulation: `functional.note_declined` records
every fused gfx950 path that was NOT taken (said once under UPP_VERBOSE=1); after one eager training step of each recipe the record
must be exactly the documented list below -- empty -- and the torch profiler must show no library GEMM (`Cijk_*`) kernel."""
import pytest
import torch

import _seeded
from models import build_model_from_cfg
from utils.config import builtin_cfg
from upp_hip import functional as HF
from upp_hip.train import freeze_for_peft, PEFT_STAGE1

pytestmark = pytest.mark.gpu

SEG_PEFT = ['downstream_adapter', 'downstream_prompts', 'label_conv', 'propagation_0', 'seg_head', 'propagation_1']   # reference tools/runner_unify_seg.py:143-146
DOCUMENTED_DECLINES = set()        # (site, reason) pairs that are allowed to run on torch: none


def _profile(step):
    step()                                              # warm-up: lazy caches (transposed frozen weights)
    HF._declined.clear()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    return set(HF._declined), [e.key for e in prof.key_averages() if e.key.startswith('Cijk')]


def test_headline_step_declines_nothing_and_launches_no_library_gemm():
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda().train()
    freeze_for_peft(m, PEFT_STAGE1)
    x = _seeded.noisy_clouds(4, 1024, seed=0).cuda()
    y = torch.tensor([1, 2, 3, 4], device='cuda')

    def step():
        for p in m.parameters():
            p.grad = None
        loss, _ = m.get_loss_acc(m(x, completion_prompt=True, denoise=True, point_num=1024), y)
        loss.backward()
    declined, gemms = _profile(step)
    assert declined == DOCUMENTED_DECLINES, declined
    assert not gemms, gemms


def test_segmentation_step_declines_nothing_and_launches_no_library_gemm():
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_shapenetpart_seg').model)).cuda().train()
    freeze_for_peft(m, SEG_PEFT)
    B = 32                                              # the benched batch: 65,536 point rows, the tall kernels
    pts = _seeded.noisy_clouds(B, 1552, seed=1).cuda()
    lpts = _seeded.unit_ball_clouds(B, 2048, seed=2).cuda()
    onehot = torch.zeros(B, 16, device='cuda'); onehot[:, 5] = 1
    tgt = torch.randint(0, 50, (B * 2048,), device='cuda')

    def step():
        for p in m.parameters():
            p.grad = None
        logp = m(pts, onehot, label_points=lpts, completion_prompt=True, denoise=True, point_num=1536)
        m.get_loss(logp.reshape(-1, 50), tgt).backward()
    declined, gemms = _profile(step)
    assert declined == DOCUMENTED_DECLINES, declined
    assert not gemms, gemms


STAGE2_KEYS = ['downstream_adapter', 'downstream_adapter1', 'downstream_prompts', 'dense_pred', 'mask_token', 'rectify_prompter',
               'shape_pred', 'coarse_pred', 'predict_token_generator', 'mask_prompter', 'mask_token_generator']   # reference tools/runner_module.py:232-238


def test_stage2_step_declines_nothing_and_launches_no_library_gemm():
    """Stage 2 of the recipe: the gradient runs through the prompted geometry (propagation weights, the 3 -> 128 position layers, the
    64 -> 3 score head, the patch embedding's data gradients) -- all on own kernels since round 3."""
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda().train()
    freeze_for_peft(m, STAGE2_KEYS)
    x = _seeded.noisy_clouds(4, 1024, seed=0).cuda()
    y = torch.tensor([1, 2, 3, 4], device='cuda')

    def step():
        for p in m.parameters():
            p.grad = None
        loss, _ = m.get_loss_acc(m(x, completion_prompt=True, denoise=True, point_num=1024), y)
        loss.backward()
    declined, gemms = _profile(step)
    assert declined == DOCUMENTED_DECLINES, declined
    assert not gemms, gemms
