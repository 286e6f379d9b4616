"""Which stage-2 gradients differ from the reference fixture on the GPU (diagnostic for tests/test_gpu_model.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "iccv2025-upp_amd")]
import numpy as np, torch
import _seeded
from models import build_model_from_cfg
from utils.config import builtin_cfg
KEYS = ['downstream_adapter', 'downstream_adapter1', 'downstream_prompts', 'dense_pred', 'mask_token', 'rectify_prompter',
        'shape_pred', 'coarse_pred', 'predict_token_generator', 'mask_prompter', 'mask_token_generator']
g = np.load(os.path.join(ROOT, "tests/golden/upp_stage2.npz"))
m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).eval().cuda()
for n, p in m.named_parameters():
    p.requires_grad_(any(k in n for k in KEYS))
os.environ["UPP_VERBOSE"] = "1"
logits = m(_seeded.noisy_clouds(2, 1024, 0).cuda(), completion_prompt=True, denoise=True, point_num=1024)
loss, _ = m.get_loss_acc(logits, torch.from_numpy(g['labels']).cuda())
loss.backward()
grads = {n: p.grad for n, p in m.named_parameters() if p.requires_grad and p.grad is not None}
for n, ref in zip(g['grad_names'], g['grad_norms']):
    got = grads[n].norm().item()
    if abs(got - ref) > 2e-3 * ref:
        print("%-60s got %.6f ref %.6f ratio %.4f" % (n, got, ref, got / ref))
