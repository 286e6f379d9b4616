"""How the f32 evaluations of the part-segmentation training step sit from the float64 reference (tests/golden/upp_seg_train_f64.npz):
relative L2, median and 99th-percentile element error (of the array's scale) -- this repository's HIP path and the reference's own f32
fixture.  ReLU-gate flips move the L2 figure by draws; the median does not see them.   python tools/micro/seg_flip_noise.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("oracle", "tests", "iccv2025-upp_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import _seeded
from models import build_model_from_cfg
from utils.config import builtin_cfg
import test_seg_golden as TS

G = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))
ex, rf, g0 = G("upp_seg_train_f64"), G("upp_seg_train"), G("upp_seg")
m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_shapenetpart_seg').model)).cuda()
TS._deterministic_train(m)
for n, p in m.named_parameters():
    p.requires_grad_(any(k in n for k in TS.SEG_PEFT))
spts, lpts = TS._inputs()
logp = m(spts.cuda(), torch.from_numpy(g0['onehot']).cuda(), label_points=lpts.cuda(), completion_prompt=True, denoise=True, point_num=1536)
m.get_loss(logp.reshape(-1, 50), torch.from_numpy(g0['target']).reshape(-1).cuda()).backward()
grads = {n: p.grad for n, p in m.named_parameters() if p.requires_grad and p.grad is not None}
def stats(a, e):
    d = np.abs(a.astype(np.float64) - e).ravel() / np.abs(e).max()
    return np.linalg.norm((a - e).ravel()) / np.linalg.norm(e.ravel()), np.median(d), np.percentile(d, 99)
for k in ex.files:
    if '::' in k and 'mlp_convs.1.bias' not in k:
        name = k.split('::', 1)[1]
        got = grads[name].cpu().numpy() if k.startswith('grad::') else grads[name].squeeze(-1)[::16, ::16].cpu().numpy()
        a, b = stats(got, ex[k]), stats(rf[k], ex[k])
        print("%-46s ours L2 %.1e med %.1e p99 %.1e | ref-f32 L2 %.1e med %.1e p99 %.1e" % ((name,) + a + b))
