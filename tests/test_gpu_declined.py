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
    names = [e.key for e in prof.key_averages()]
    LAST_KERNELS[:] = names
    return set(HF._declined), [k for k in names if k.startswith('Cijk')]


LAST_KERNELS = []          # kernel / op names of the last profiled step


def _library_sorts():
    return [k for k in LAST_KERNELS if 'rocprim' in k.lower() or 'radixsort' in k.lower() or 'segmented_sort' in k.lower()]


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
    assert not _library_sorts(), _library_sorts()


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
    assert not _library_sorts(), _library_sorts()


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
    assert not _library_sorts(), _library_sorts()


# ---- the secondary recipes (round-4 verdict: the same assertions for pretask, pretrain and cls_aux, plus: no library sort) ------------
PRETASK_PEFT = ['rectify_adapter', 'downstream_adapter', 'pretask_adapter', 'rectify_prompts', 'downstream_prompts', 'pretask_prompts',
                'coarse_pred', 'increase_dim', 'mask_token', 'dense_pred', 'rectify_prompter', 'shape_pred', 'predict_token_generator',
                'mask_prompter', 'mask_token_generator']                                   # reference tools/runner_pretask.py:112-117


def test_pretask_step_declines_nothing_and_launches_no_library_gemm_or_sort():
    """Pre-task recipe (reference tools/runner_pretask.py:157-247, models/Point_MAE_pretask_dev.py:655-741): three Chamfer-L1 terms + the
    noise loss; the ranking of the noise recall is the rank-counting kernel (upp_argsort_rows), not a radix sort."""
    from models.Point_MAE_pretask_dev import pretask_losses
    from utils import misc
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('pretask').model)).cuda().train()
    freeze_for_peft(m, PRETASK_PEFT)
    B = 4
    g = torch.Generator(device='cuda').manual_seed(0)
    gt = _seeded.unit_ball_clouds(B, 8192, seed=1).cuda()
    partial, cropping = misc.seprate_point_cloud(gt, 8192, 2048, sample_points=1024, incomplete_shape=True, generator=g)
    noise = [misc.gaussian_noise([B, 20, 3], loc=0., scale=0.2, shell_radius=0.8, device='cuda', generator=g),
             misc.lidar_noise(partial, 32, low=1.2, scale=1.5, generator=g)]
    points = torch.cat([partial] + noise, dim=1).contiguous()

    def step():
        for p in m.parameters():
            p.grad = None
        total, _ = pretask_losses(m, gt, partial, cropping, points, point_num=1024)
        total.backward()
    declined, gemms = _profile(step)
    assert declined == DOCUMENTED_DECLINES, declined
    assert not gemms, gemms
    assert not _library_sorts(), _library_sorts()


def test_pretrain_step_declines_nothing_and_launches_no_library_gemm_or_sort():
    """Point-MAE pre-training (reference tools/runner_pretrain.py:115-148, models/Point_MAE.py:277-329): every parameter trainable; the
    masking orders (argsort of uniform draws, the visible-first order) on the rank-counting kernel."""
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('pretrain').model)).cuda().train()
    x = _seeded.unit_ball_clouds(4, 1024, seed=1).cuda()

    def step():
        for p in m.parameters():
            p.grad = None
        m(x).backward()
    declined, gemms = _profile(step)
    assert declined == DOCUMENTED_DECLINES, declined
    assert not gemms, gemms
    assert not _library_sorts(), _library_sorts()


def test_cls_aux_step_declines_nothing_and_launches_no_library_gemm_or_sort():
    """The headline step + ChamferDistanceL1 + EMD on the completion prompter's rebuilt cloud (BASELINE configs[2] wording)."""
    from extensions.chamfer_dist import ChamferDistanceL1
    from emd import emd
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda().train()
    freeze_for_peft(m, PEFT_STAGE1)
    x = _seeded.noisy_clouds(4, 1024, seed=0).cuda()
    gt = _seeded.unit_ball_clouds(4, 1024, seed=0).cuda()
    y = torch.tensor([1, 2, 3, 4], device='cuda')
    cd_l1, emd_loss = ChamferDistanceL1(), emd()

    def step():
        for p in m.parameters():
            p.grad = None
        ce, _ = m.get_loss_acc(m(x, completion_prompt=True, denoise=True, point_num=1024), y)
        rebuild = m.aux['rebuild_points'].detach().requires_grad_(True)
        (ce + cd_l1(rebuild, gt) + emd_loss(rebuild, gt)).backward()
    declined, gemms = _profile(step)
    assert declined == DOCUMENTED_DECLINES, declined
    assert not gemms, gemms
    assert not _library_sorts(), _library_sorts()
