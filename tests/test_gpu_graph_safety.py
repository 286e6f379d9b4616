"""A captured step must contain no memset node.  On this stack (ROCm 7.2.0 / torch 2.10.0+rocm7.0) a memset NODE of a HIP graph works in
the first replay and writes garbage from the second on (measured: tools/micro/memset_graph_check.py; NOTEBOOK 12.11).  torch reductions
that split their rows over workgroups zero their semaphores with cudaMemsetAsync; this library used hipMemsetAsync in two fallback
paths.  Round 6 removed both kinds from every recipe's step (own column-sum kernels, a zero-fill kernel); these tests keep it that way:
the torch profiler lists the memsets of one eager forward + backward + update of every shipped recipe (what a capture would turn into
nodes), and the two entry points that used to memset are replayed from a graph against their eager results."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,batch", [("cls", 32), ("cls_aux", 32), ("stage2", 32), ("pretask", 32), ("pretrain", 32), ("seg", 32), ("seg", 4), ("cls", 4)])
def test_a_step_issues_no_memset(kind, batch):
    import bench
    from memset_census import memsets_of
    dev = torch.device("cuda", 0)
    tr = bench.Trainer(dev, batch, False, use_graph=False) if kind == "cls" else bench.RecipeTrainer(kind, dev, batch, use_graph=False)
    for _ in range(2):
        tr.ts._forward_backward()
        tr.ts._update()
    torch.cuda.synchronize()

    def step():
        tr.ts._forward_backward()
        tr.ts._update()
    found = memsets_of(step)
    assert not found, "memsets in the %s step (memset nodes once captured):\n%s" % (kind, "\n".join("  %s %s | %s" % f for f in found))


@pytest.mark.parametrize("kind", ["cls_aux", "stage2", "pretask", "pretrain", "seg"])
def test_replayed_steps_stay_finite_and_follow_the_eager_loss(kind):
    """Six replays of every secondary recipe's captured step: loss and gradient buffer finite in every one (a memset node that
    misfired wrote -1.15e37 into the EMD cost and inf into Chamfer gradients from the SECOND replay on), and the first losses within
    the spread of the eager driver's (same weights, same batches; dropout and masks draw different numbers)."""
    import bench
    dev = torch.device("cuda", 0)
    B = 8
    losses = {}
    for use_graph in (False, True):
        torch.manual_seed(7)
        tr = bench.RecipeTrainer(kind, dev, B, use_graph=use_graph, pipeline=False)
        out = []
        for _ in range(6):
            loss = tr.step()
            torch.cuda.synchronize()
            out.append(float(loss))
            assert torch.isfinite(tr.ts.flat.flat).all() and float(tr.ts.flat.flat.abs().max()) > 0.0, (kind, use_graph, out)
        assert all(v == v and abs(v) < 1e6 for v in out), (kind, use_graph, out)
        losses[use_graph] = out
    e, g = losses[False], losses[True]
    assert abs(sum(g) / len(g) - sum(e) / len(e)) <= 0.25 * abs(sum(e) / len(e)) + 0.05, (kind, e, g)


@pytest.mark.parametrize("kind,batch", [("seg", 4), ("seg", 8), ("cls_aux", 8), ("stage2", 8), ("pretask", 8)])
def test_replays_equal_the_eager_steps(kind, batch):
    """With every random draw switched off (dropout and stochastic depth at 0) a step is a function of the weights and the batch: the
    captured step's replays must walk the eager driver's trajectory -- loss and gradient buffer of steps 1 ... 4, which covers the
    first replay, the second (where a memset node first misfired) and two more.  Segmentation at B = 4 takes the small-batch forms
    (per-group bias as an own node, GEMMs below the tall-tile limit), at B = 8 the tall ones.  (The pre-training recipe draws its mask.)"""
    import bench
    dev = torch.device("cuda", 0)
    runs = {}
    for use_graph in (False, True):
        tr = bench.RecipeTrainer(kind, dev, batch, use_graph=use_graph, pipeline=False)
        for m in tr.model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if hasattr(m, "drop_prob"):
                m.drop_prob = 0.0
        out = []
        for _ in range(4):
            loss = tr.step()
            torch.cuda.synchronize()
            out.append((float(loss), tr.ts.flat.flat.clone()))
        runs[use_graph] = out
    # (stage 2 and the pre-task recipe differentiate through atomic-add kernels -- the interpolation's geometry gradient, the Chamfer
    #  gradient of the 2,048 x 8,192 pair: the order of the adds differs from run to run, and AdamW carries the last bits forward)
    #  -- and there the comparison stops after the second step: from the third on two EAGER runs part company too, a discrete choice of
    #  the forward (FPS picks on the completed cloud) flips on the carried bits: tools/micro/replay_vs_eager.py)
    loose = kind in ("stage2", "pretask")
    tol = 2e-3 if loose else 2e-5
    for k, ((le, ge), (lg, gg)) in enumerate(list(zip(runs[False], runs[True]))[:2 if loose else 4]):
        assert abs(le - lg) <= max(2e-6, 0.1 * tol) * abs(le), (k, le, lg)
        scale = float(ge.abs().max())
        assert float((ge - gg).abs().max()) <= tol * scale, (k, float((ge - gg).abs().max()), scale)


def test_the_profiler_sees_a_memset_when_there_is_one():
    """(the check above must not pass because the profiler is blind: torch's own multi-workgroup reduction has one)"""
    from memset_census import memsets_of
    x = torch.randn(4, 2048, 512, device='cuda')
    x.sum(1)
    torch.cuda.synchronize()
    assert memsets_of(lambda: x.sum(1))


def test_entry_points_that_used_to_memset_replay_correctly_from_a_graph():
    from upp_hip import ops
    torch.manual_seed(0)
    B, n, m = 4, 256, 256
    x1, x2 = torch.rand(B, n, 3, device='cuda'), torch.rand(B, m, 3, device='cuda')
    match = ops.emd_approxmatch(x1, x2)
    cost = ops.emd_matchcost(x1, x2, match).clone()
    y1, y2 = torch.rand(2, 2048, 3, device='cuda'), torch.rand(2, 8192, 3, device='cuda')         # n + m > 5,461: the atomic-add path
    d1, d2, i1, i2 = ops.chamfer_fwd(y1, y2)
    gd1, gd2 = torch.rand_like(d1), torch.rand_like(d2)
    e1, e2 = [t.clone() for t in ops.chamfer_bwd(y1, y2, i1, i2, gd1, gd2)]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.emd_matchcost(x1, x2, match)
        ops.chamfer_bwd(y1, y2, i1, i2, gd1, gd2)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        c = ops.emd_matchcost(x1, x2, match)
        g1, g2 = ops.chamfer_bwd(y1, y2, i1, i2, gd1, gd2)
    for _ in range(4):
        g.replay()
        torch.cuda.synchronize()
        assert torch.allclose(c, cost, rtol=1e-5, atol=0)                                 # (atomic adds: order, not bits)
        assert torch.allclose(g1, e1, rtol=1e-4, atol=1e-6) and torch.allclose(g2, e2, rtol=1e-4, atol=1e-6)


def test_own_column_sums_replace_torch_reductions():
    from upp_hip import ops
    import upp_hip.functional as HF
    torch.manual_seed(1)
    for rows, cols, off, length in ((130, 768, 0, 384), (1, 40, 4, 36), (5000, 96, 0, 96), (70000, 64, 8, 40)):
        x = torch.randn(rows, cols, device='cuda')
        ref = x[:, off:off + length].double().sum(0)
        got = ops.sum_rows(x, off, length)
        assert got.shape == (length,) and float((got.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())) * (rows ** 0.5)
    tok = torch.randn(1, 1, 384, device='cuda', requires_grad=True)
    w = torch.randn(8, 152, 384, device='cuda')
    (gt,) = torch.autograd.grad((HF.expand_rows(tok, 8, 152) * w).sum(), [tok])
    assert gt.shape == tok.shape and torch.allclose(gt.view(-1), w.view(-1, 384).sum(0), rtol=1e-5, atol=1e-4)
    y = torch.randn(4 * 2048, 512, device='cuda', requires_grad=True)
    gb = torch.randn(4, 512, device='cuda', requires_grad=True)
    wg = torch.randn(4 * 2048, 512, device='cuda')
    out = HF._GroupBiasAdd.apply(y, gb, 2048)
    assert torch.equal(out, (y.view(4, 2048, 512) + gb.unsqueeze(1)).view(-1, 512))
    gy, ggb = torch.autograd.grad(out, [y, gb], grad_outputs=wg)
    assert torch.equal(gy, wg) and torch.allclose(ggb, wg.view(4, 2048, 512).sum(1), rtol=1e-5, atol=1e-3)


def test_memcpy_nodes_replay_correctly():
    """(the step drivers feed their static input buffers with device-to-device copies, some of them memcpy NODES of a graph: unlike memset
    nodes they replay correctly -- checked here so that a runtime that changes this is noticed)"""
    src = torch.zeros(1 << 20, device='cuda')
    dst = torch.empty_like(src)
    small_src, small_dst = torch.zeros(3, device='cuda'), torch.empty(3, device='cuda')
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dst.copy_(src)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        dst.copy_(src)
        small_dst.copy_(small_src)
    for it in range(4):
        src.fill_(float(it + 1))
        small_src.fill_(float(10 * it + 1))
        g.replay()
        torch.cuda.synchronize()
        assert float(dst.min()) == float(dst.max()) == float(it + 1)
        assert small_dst.tolist() == [float(10 * it + 1)] * 3
