"""Where does a pipelined segmentation step diverge from its serialised self?  Runs the set-up of
tests/test_gpu_block.py::test_pipelined_step_with_a_recipe_back_end_equals_the_sequential_order several times, once with the two halves
serialised (UPP_PIPE_SERIAL) as the reference, and reports per step which hand-over tensors / loss / gradient entries differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("oracle", "tests", "iccv2025-upp_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
os.environ["UPP_DIAG_AUX"] = "1"
import torch
import _seeded
from models import build_model_from_cfg
from utils.config import builtin_cfg
from upp_hip.train import PipelinedTrainStep, freeze_for_peft

keys = ['downstream_adapter', 'downstream_prompts', 'bnorm', 'label_conv', 'propagation_0', 'seg_head']
B, steps = 2, 4
raws = [torch.cat([_seeded.noisy_clouds(B, 1536, seed=60 + k), _seeded.unit_ball_clouds(B, 16, seed=70 + k) * 1.01], 1).contiguous().cuda() for k in range(steps)]
lpts = [_seeded.unit_ball_clouds(B, 2048, seed=80 + k).cuda() for k in range(steps)]
onehot = torch.zeros(B, 16, device='cuda'); onehot[torch.arange(B), torch.arange(B) % 16] = 1
g = torch.Generator(device='cuda').manual_seed(3)
targets = [torch.randint(0, 50, (B * 2048,), device='cuda', generator=g) for _ in range(steps)]

def front_fn(m, x): return m.prompt_tokens(x, True, True, 1536)
def back_fn(m, state, onehot, lp, target):
    loss = m.get_loss(m.forward_tokens(state, onehot, lp).reshape(-1, 50), target)
    return loss, loss.detach()

def make():
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_shapenetpart_seg').model)).cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
        if hasattr(mod, 'drop_prob'): mod.drop_prob = 0.0
    freeze_for_peft(m, keys)
    return m

def run(serial):
    if serial: os.environ["UPP_PIPE_SERIAL"] = "1"
    else: os.environ.pop("UPP_PIPE_SERIAL", None)
    m = make()
    pipe = PipelinedTrainStep(m, tuple(raws[0].shape), forward_kwargs=dict(completion_prompt=True, denoise=True, point_num=1536),
                              front_fn=front_fn, back_fn=back_fn, extras=[onehot, lpts[0], targets[0]], back_end_keys=tuple(keys), lr=0.0)
    pipe._capture()
    rec = []
    for k in range(steps):
        pipe.step(raws[k], extras=[onehot, lpts[k], targets[k]])
        torch.cuda.synchronize()
        p = k & 1
        rec.append(dict(state=[m.aux[n].clone() for n in ('dbg_rectified', 'dbg_vis_center', 'rebuild_points', 'dbg_sampled')] + [t.clone() for t in pipe.state[p]], loss=float(pipe.loss) if k > 0 else None, grad=pipe.flat.flat.clone() if k > 0 else None))
    pipe.flush(); torch.cuda.synchronize()
    rec.append(dict(state=[], loss=float(pipe.loss), grad=pipe.flat.flat.clone()))
    return rec

ref = run(True)
ref2 = run(True)
def diff(a, b, tag):
    out = []
    for k, (x, y) in enumerate(zip(a, b)):
        for i, (s, t) in enumerate(zip(x['state'], y['state'])):
            if not torch.equal(s, t) and s.dim() == 3 and s.shape[-1] == 3:
                bad = (s != t).any(-1)
                for b in range(s.shape[0]):
                    idx = bad[b].nonzero().flatten().tolist()
                    if idx:
                        same_set = sorted(map(tuple, s[b].tolist())) == sorted(map(tuple, t[b].tolist()))
                        out.append("   sample %d: %d points differ, first %s last %s; same point SET: %s" % (b, len(idx), idx[:6], idx[-3:], same_set))
            if not torch.equal(s, t): out.append("step %d state[%d] %s max|d| %.3g (%d entries)" % (k, i, tuple(s.shape), (s - t).abs().max().item(), (s != t).sum().item()))
        if x['loss'] is not None and x['loss'] != y['loss']: out.append("step %d loss %.8f vs %.8f" % (k, x['loss'], y['loss']))
        if x['grad'] is not None and not torch.equal(x['grad'], y['grad']):
            d = (x['grad'] - y['grad']).abs(); out.append("step %d grad max|d| %.3g of scale %.3g (%d entries of %d)" % (k, d.max().item(), y['grad'].abs().max().item(), (d > 0).sum().item(), d.numel()))
    print(tag, "IDENTICAL" if not out else "\n   " + "\n   ".join(out), flush=True)
diff(ref2, ref, "serial vs serial:")
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    diff(run(False), ref, "pipelined run %d vs serial:" % i)
