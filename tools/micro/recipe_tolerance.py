"""Measured deviation of the GPU evaluation of the secondary recipes from their reference fixtures (what the tolerances of
tests/test_point_mae.py / tests/test_pretask.py are set from): max relative deviation of the loss, of the gradient norms and of the
gradient arrays the fixtures hold.   python tools/micro/recipe_tolerance.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("oracle", "tests", "iccv2025-upp_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import _seeded
from models import build_model_from_cfg
from models.Point_MAE_pretask_dev import pretask_losses
from utils.config import builtin_cfg
G = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))

g = G("point_mae")
m = _seeded.fill(build_model_from_cfg(builtin_cfg('pretrain').model)).eval().cuda()
pts = _seeded.unit_ball_clouds(2, 1024, seed=21).cuda()
loss = m(pts, mask=torch.from_numpy(g['mask']).cuda()); loss.backward()
grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
norms = np.array([grads[n].norm().item() for n in g['grad_names']])
print("point_mae  loss rel %.2e | norms max rel %.2e | mask_token max abs/scale %.2e | increase_bias %.2e" % (
    abs(loss.item() / g['loss'] - 1), np.abs(norms / g['grad_norms'] - 1).max(),
    np.abs(grads['mask_token'].cpu().numpy() - g['g_mask_token']).max() / np.abs(g['g_mask_token']).max(),
    np.abs(grads['increase_dim.0.bias'].cpu().numpy() - g['g_increase_bias']).max() / np.abs(g['g_increase_bias']).max()))

g = G("pretask")
m = _seeded.fill(build_model_from_cfg(builtin_cfg('pretask').model)).cuda()
m.train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
    if hasattr(mod, 'drop_prob'): mod.drop_prob = 0.0
gt = _seeded.unit_ball_clouds(2, 1280, seed=31)
partial, cropping = gt[:, :1024].contiguous(), gt[:, 1024:].contiguous()
noise = _seeded.noisy_clouds(2, 1024, seed=32)[:, 1024:1076].contiguous()
points = torch.cat([partial, noise], dim=1).contiguous()
total, terms = pretask_losses(m, gt.cuda(), partial.cuda(), cropping.cuda(), points.cuda(), point_num=1024)
total.backward()
grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
norms = np.array([grads[n].norm().item() for n in g['grad_names']])
big = g['grad_norms'] > 1e-4
print("pretask    loss rel %.2e | terms %s | norms max rel (norm > 1e-4) %.2e, max abs (others) %.2e" % (
    abs(total.item() / g['loss'] - 1), {k: "%.1e" % abs(terms[t].item() / g[k] - 1) for k, t in (('coarse', 'cropping_coarse'), ('crop_dense', 'cropping_dense'), ('dense', 'dense'), ('noise_loss', 'noise'))},
    np.abs(norms[big] / g['grad_norms'][big] - 1).max(), np.abs(norms[~big] - g['grad_norms'][~big]).max()))

# ---- headline PEFT step and stage 2 (tests/test_gpu_model.py): loss, gradient norms, gradient arrays of the reference-class fixtures
PEFT_KEYS = ['downstream_adapter', 'downstream_adapter1', 'downstream_prompts', 'bnorm', 'cls_pos', 'cls_token', 'cls_head_finetune']
STAGE2_KEYS = ['downstream_adapter', 'downstream_adapter1', 'downstream_prompts', 'dense_pred', 'mask_token', 'rectify_prompter',
               'shape_pred', 'coarse_pred', 'predict_token_generator', 'mask_prompter', 'mask_token_generator']
m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).eval().cuda()
for name, keys in (("upp_model", PEFT_KEYS), ("upp_stage2", STAGE2_KEYS)):
    g = G(name)
    for n, p in m.named_parameters():
        p.requires_grad_(any(k in n for k in keys)); p.grad = None
    logits = m(_seeded.noisy_clouds(2, 1024, 0).cuda(), completion_prompt=True, denoise=True, point_num=1024)
    loss, _ = m.get_loss_acc(logits, torch.from_numpy(g['labels']).cuda())
    loss.backward()
    grads = {n: p.grad for n, p in m.named_parameters() if p.requires_grad and p.grad is not None}
    norms = np.array([grads[n].norm().item() for n in g['grad_names']])
    worst_abs, worst_rel, worst_l2 = 0.0, 0.0, 0.0
    for k in g.files:
        if k.startswith('grad::'):
            ref = g[k]; got = grads[k[6:]].cpu().numpy()
            scale = np.abs(ref).max()
            worst_abs = max(worst_abs, np.abs(got - ref).max() / scale)
            worst_l2 = max(worst_l2, np.linalg.norm(got - ref) / np.linalg.norm(ref))
            sig = np.abs(ref) > 1e-2 * scale
            worst_rel = max(worst_rel, (np.abs(got - ref)[sig] / np.abs(ref)[sig]).max())
    print("%-10s loss rel %.2e | norms max rel %.2e | arrays: max abs/scale %.2e, max rel (|ref| > 1e-2 scale) %.2e, rel L2 %.2e" % (
        name, abs(loss.item() / g['loss'] - 1), np.abs(norms / g['grad_norms'] - 1).max(), worst_abs, worst_rel, worst_l2))
    if name == "upp_stage2":
        for k in g.files:
            if k.startswith('grad::'):
                ref = g[k]; got = grads[k[6:]].cpu().numpy()
                print("    %-60s max abs/scale %.2e  rel L2 %.2e  |ref|max %.2e" % (k[6:], np.abs(got - ref).max() / np.abs(ref).max(), np.linalg.norm(got - ref) / np.linalg.norm(ref), np.abs(ref).max()))
        bad = [(abs(a / b - 1), n) for n, a, b in zip(g['grad_names'], norms, g['grad_norms']) if abs(a / b - 1) > 5e-5]
        for e, n in sorted(bad, reverse=True)[:12]:
            print("    norm %-60s rel %.2e" % (n, e))
