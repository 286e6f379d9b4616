"""Stress the two-stream pipelined step for hand-over races: N steps with a different batch each step at lr = 0, against the
same calls run one after the other; per-step losses and the back-end BatchNorm statistics (order-sensitive functions of every
hand-over) must agree.   python tools/pipeline_stress.py [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "iccv2025-upp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import _seeded  # noqa: E402
from models import build_model_from_cfg  # noqa: E402
from utils.config import builtin_cfg  # noqa: E402
from upp_hip.train import TrainStep, PipelinedTrainStep, freeze_for_peft  # noqa: E402


def make():
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, 'drop_prob'):
            mod.drop_prob = 0.0
    freeze_for_peft(m)
    return m


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    B = 8
    base = [_seeded.noisy_clouds(B, 1024, seed=500 + k).cuda() for k in range(16)]
    g = torch.Generator().manual_seed(1)
    perm = [torch.randperm(1096, generator=g).cuda() for _ in range(steps)]
    labels = [torch.randint(0, 40, (B,), generator=g).cuda() for _ in range(steps)]

    def batch(k):
        return base[k % 16][:, perm[k]].contiguous() * (1.0 + 0.001 * (k % 7))

    kw = dict(completion_prompt=True, denoise=True, point_num=1024)
    kw_back = dict(kw, completion_prompt=False, denoise=False)
    m_ref = make()
    ref = TrainStep(m_ref, (B, 1096, 3), use_graph=False, forward_kwargs=kw, lr=0.0)
    losses_ref = []
    for k in range(steps):
        with torch.no_grad():
            st = m_ref.prompt_tokens(batch(k), True, True, 1024)
        ref._forward_backward(st, labels[k], kw_back)
        ref._update()
        losses_ref.append(float(ref.loss))

    m_pipe = make()
    pipe = PipelinedTrainStep(m_pipe, (B, 1096, 3), forward_kwargs=kw, lr=0.0)
    pipe._capture()
    _seeded.fill(m_pipe)
    losses = []
    for k in range(steps):
        pipe.step(batch(k), labels[k])
        if k > 0:
            losses.append(float(pipe.loss) if k % 16 == 0 else pipe.loss.clone())      # mostly no host sync inside the loop
    pipe.flush()
    losses.append(float(pipe.loss))
    losses = [float(x) for x in losses]
    np.testing.assert_allclose(losses, losses_ref, rtol=1e-4)
    sd_ref, sd = m_ref.state_dict(), m_pipe.state_dict()
    worst = 0.0
    for kname in sd_ref:
        if 'running_' in kname and not kname.startswith('encoder.'):
            a, b = sd[kname].double(), sd_ref[kname].double()
            worst = max(worst, ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item())
    assert worst < 1e-4, worst
    print("pipeline stress: %d steps, losses agree (max rel diff %.2e), BatchNorm statistics agree (%.2e)" % (
        steps, max(abs(a - b) / abs(b) for a, b in zip(losses, losses_ref)), worst))


if __name__ == "__main__":
    main()
