"""The pipelined step driver (front-end of batch k+1 on a second stream beside forward + backward of batch k) is DETERMINISTIC: over 24
steps with a real learning rate it hands over the bits its serialised self hands over (UPP_PIPE_SERIAL=1: same graphs, the halves one
after the other) -- every front-end state tensor of every step (FPS picks decide them: centres, index lists, tokens), every loss, the
gradient buffer and the final parameters.  Round 4 found 3 of 10 pipelined segmentation runs diverging (v_pk_add_f32 with op_sel beside
a bf16-MFMA workgroup changed FPS picks: DESIGN section 4) with a probe, not a test; this is the test (round-4 verdict, item 4).  Headline
recipe (BASELINE configs[2]) and part segmentation (configs[4]); reference: tools/runner_module.py:193-212, runner_unify_seg.py."""
import os

import pytest
import torch

import _seeded
from models import build_model_from_cfg, upp_layers
from utils.config import builtin_cfg
from upp_hip.train import PipelinedTrainStep, freeze_for_peft, PEFT_STAGE1

pytestmark = pytest.mark.gpu
STEPS = 24


def _run(make_pipe, feed, serial):
    if serial:
        os.environ["UPP_PIPE_SERIAL"] = "1"
    else:
        os.environ.pop("UPP_PIPE_SERIAL", None)
    try:
        torch.manual_seed(1234)                      # (the drivers derive their front-end generator from the global seed)
        # the per-forward uniform bank sizes itself from the previous forward: a run that starts with an empty bank draws its first
        # forward's uniforms one by one, a later run in one launch -- other values from the same seed.  Every run starts from an empty bank.
        bank = upp_layers.UNIFORMS
        bank.buf, bank.pos, bank.asked, bank.need = None, 0, 0, 0
        pipe, model = make_pipe()
        pipe._capture()
        rec = []
        for k in range(STEPS):
            feed(pipe, k)
            torch.cuda.synchronize()
            rec.append(([t.clone() for t in pipe.state[k & 1]], pipe.loss.clone(), pipe.flat.flat.clone()))
        pipe.flush()
        torch.cuda.synchronize()
        params = {n: v.detach().clone() for n, v in model.state_dict().items()}
        return rec, pipe.loss.clone(), params
    finally:
        os.environ.pop("UPP_PIPE_SERIAL", None)


def _assert_identical(a, b):
    (rec_a, loss_a, par_a), (rec_b, loss_b, par_b) = a, b
    for k, ((sa, la, ga), (sb, lb, gb)) in enumerate(zip(rec_a, rec_b)):
        for i, (x, y) in enumerate(zip(sa, sb)):
            assert torch.equal(x, y), "step %d: hand-over tensor %d %s differs in %d entries" % (k, i, tuple(x.shape), int((x != y).sum()))
        assert torch.equal(la, lb), "step %d: loss %r vs %r" % (k, float(la), float(lb))
        assert torch.equal(ga, gb), "step %d: gradient buffer differs in %d entries" % (k, int((ga != gb).sum()))
    assert torch.equal(loss_a, loss_b)
    for n in par_a:
        assert torch.equal(par_a[n], par_b[n]), "parameter / buffer %s differs after %d steps" % (n, STEPS)


def test_headline_pipelined_step_is_bit_identical_to_its_serialised_self_over_24_steps():
    B = 8
    batches = [(_seeded.noisy_clouds(B, 1024, seed=300 + k).cuda(), torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(k)).cuda())
               for k in range(4)]

    def make_pipe():
        m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda().train()
        freeze_for_peft(m, PEFT_STAGE1)
        return PipelinedTrainStep(m, tuple(batches[0][0].shape), lr=5e-4), m

    def feed(pipe, k):
        pipe.step(*batches[k & 3])
    ref = _run(make_pipe, feed, serial=True)
    _assert_identical(_run(make_pipe, feed, serial=True), ref)            # the serialised driver is reproducible at all
    for _ in range(2):
        _assert_identical(_run(make_pipe, feed, serial=False), ref)


def test_segmentation_pipelined_step_is_bit_identical_to_its_serialised_self_over_24_steps():
    keys = ['downstream_adapter', 'downstream_prompts', 'label_conv', 'propagation_0', 'seg_head', 'propagation_1']      # reference tools/runner_unify_seg.py:143-146
    B = 4
    raws = [torch.cat([_seeded.noisy_clouds(B, 1536, seed=60 + k), _seeded.unit_ball_clouds(B, 16, seed=70 + k) * 1.01], 1).contiguous().cuda() for k in range(4)]
    lpts = [_seeded.unit_ball_clouds(B, 2048, seed=80 + k).cuda() for k in range(4)]
    onehot = torch.zeros(B, 16, device='cuda')
    onehot[torch.arange(B), torch.arange(B) % 16] = 1
    g = torch.Generator(device='cuda').manual_seed(3)
    targets = [torch.randint(0, 50, (B * 2048,), device='cuda', generator=g) for _ in range(4)]

    def front_fn(m, x):
        return m.prompt_tokens(x, True, True, 1536)

    def back_fn(m, state, onehot, lp, target):
        loss = m.get_loss(m.forward_tokens(state, onehot, lp).reshape(-1, 50), target)
        return loss, loss.detach()

    def make_pipe():
        m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_shapenetpart_seg').model)).cuda().train()
        freeze_for_peft(m, keys)
        pipe = PipelinedTrainStep(m, tuple(raws[0].shape), forward_kwargs=dict(completion_prompt=True, denoise=True, point_num=1536),
                                  front_fn=front_fn, back_fn=back_fn, extras=[onehot, lpts[0], targets[0]], back_end_keys=tuple(keys), lr=5e-4)
        return pipe, m

    def feed(pipe, k):
        pipe.step(raws[k & 3], extras=[onehot, lpts[k & 3], targets[k & 3]])
    ref = _run(make_pipe, feed, serial=True)
    for _ in range(2):
        _assert_identical(_run(make_pipe, feed, serial=False), ref)
