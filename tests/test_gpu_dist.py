"""Two ranks sharing the one GPU of the test box (gloo moves the flat gradient buffer: the collective itself is not what
is tested): the pipelined, graph-captured step must keep every rank's parameters identical while the ranks see different
batches, and must differ from a run without the exchange."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, paths, kind="pipelined"):
    sys.path[:0] = paths
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import _seeded
    from models import build_model_from_cfg
    from upp_hip.train import PipelinedTrainStep, freeze_for_peft
    from utils import dist_utils
    from utils.config import builtin_cfg
    if world > 1:
        dist_utils.init_dist('pytorch', backend='gloo')
    torch.cuda.set_device(0)
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda().train()
    if kind == "rank_seeded":
        # every rank starts from DIFFERENT trainable weights and feeds rank-specific data BEFORE the first step (= before the
        # capture warm-up): the construction-time broadcast and the trace-free warm-up must still give identical replicas
        torch.manual_seed(1234 + rank)
        with torch.no_grad():
            for n_, p_ in m.named_parameters():
                if 'cls_head_finetune' in n_ or 'downstream_adapter' in n_:
                    p_.add_(0.01 * torch.randn_like(p_))
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, 'drop_prob'):
            mod.drop_prob = 0.0
    freeze_for_peft(m)
    if kind == "rank_seeded":
        from upp_hip.train import TrainStep
        ts = TrainStep(m, (4, 1096, 3))
        ts.pts.copy_(_seeded.noisy_clouds(4, 1024, seed=900 + rank).cuda())       # what bench.Trainer does before step()
        ts.labels.copy_(torch.tensor([rank, 5, 6, 7], device='cuda'))
        bn = m.blocks.blocks[0].bnorm
        before = (bn.running_mean.clone(), bn.num_batches_tracked.clone())
        ts._capture()
        torch.cuda.synchronize()
        assert float(ts.opt.state[0]) == 0.0, "the capture warm-up left optimizer steps behind"
        assert torch.equal(bn.running_mean, before[0]) and torch.equal(bn.num_batches_tracked, before[1])
        ts.flush = lambda: None
    elif kind == "onestream":
        from upp_hip.train import TrainStep
        ts = TrainStep(m, (4, 1096, 3))
        ts.flush = lambda: None
    else:
        ts = PipelinedTrainStep(m, (4, 1096, 3))
    assert ts.distributed == (world > 1)
    for k in range(4):
        pts = _seeded.noisy_clouds(4, 1024, seed=100 * rank + k).cuda()
        labels = torch.tensor([(rank + k) % 40, 3, 17, 39 - k], device='cuda')
        ts.step(pts, labels)
        if k == (1 if kind == "pipelined" else 0):        # the (all-reduced) gradient buffer of batch 0: the pipelined step finishes it one call later
            torch.cuda.synchronize()
            np.save(os.path.join(out_dir, "grad0_%s_w%d_r%d.npy" % (kind, world, rank)), ts.flat.flat.detach().cpu().numpy())
    ts.flush()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in ts.trainable]).cpu().numpy()
    np.save(os.path.join(out_dir, "params_%s_w%d_r%d.npy" % (kind, world, rank)), flat)
    np.save(os.path.join(out_dir, "loss_%s_w%d_r%d.npy" % (kind, world, rank)), np.array(float(ts.loss)))
    if kind == "rank_seeded":
        assert float(ts.opt.state[0]) == 4.0          # four steps, not four + warm-up
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def test_two_ranks_keep_identical_parameters(tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    paths = [here, os.path.join(root, "iccv2025-upp_amd"), os.path.join(root, "oracle")]
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), paths), nprocs=2, join=True)
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), paths), nprocs=1, join=True)
    a = np.load(tmp_path / "params_pipelined_w2_r0.npy"); b = np.load(tmp_path / "params_pipelined_w2_r1.npy")
    solo = np.load(tmp_path / "params_pipelined_w1_r0.npy")
    np.testing.assert_array_equal(a, b)                       # one all-reduce per step: bit-identical replicas
    assert np.abs(a - solo).max() > 1e-6                      # ... and the other rank's batches did contribute
    assert np.isfinite(np.load(tmp_path / "loss_pipelined_w2_r0.npy")) and np.isfinite(np.load(tmp_path / "loss_pipelined_w2_r1.npy"))


def test_pipelined_and_one_stream_steps_agree_at_world_size_two(tmp_path):
    """The pipelined step places its gradient all-reduce between the back-end graph and the optimizer graph while the next batch's
    front-end is already running (train.py PipelinedTrainStep._finish); the one-stream step reduces after its only graph.  Same batches,
    two ranks: the same parameters on every rank either way (the front-end reads no trainable parameter: order does not matter)."""
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    paths = [here, os.path.join(root, "iccv2025-upp_amd"), os.path.join(root, "oracle")]
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), paths), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), paths, "onestream"), nprocs=2, join=True)
    pipe = [np.load(tmp_path / ("params_pipelined_w2_r%d.npy" % r)) for r in range(2)]
    one = [np.load(tmp_path / ("params_onestream_w2_r%d.npy" % r)) for r in range(2)]
    np.testing.assert_array_equal(one[0], one[1])             # bit-identical replicas either way
    np.testing.assert_array_equal(pipe[0], pipe[1])
    # the reduced gradient of the first batch (same parameters on both sides): equal up to the f32 re-association between the front-end's
    # no-grad kernels (fused activation epilogues) and the one-graph forward -- 1e-4 of the gradient's scale.  (The PARAMETERS after four
    # AdamW steps are not compared element-wise: m / sqrt(v) turns a 1e-7 gradient difference at a near-zero gradient into a full +-lr step.)
    gp = [np.load(tmp_path / ("grad0_pipelined_w2_r%d.npy" % r)) for r in range(2)]
    go = [np.load(tmp_path / ("grad0_onestream_w2_r%d.npy" % r)) for r in range(2)]
    np.testing.assert_array_equal(gp[0], gp[1])
    np.testing.assert_array_equal(go[0], go[1])
    np.testing.assert_allclose(gp[0], go[0], rtol=1e-3, atol=1e-4 * np.abs(go[0]).max())
    # ... and the runs stay together: after four steps no parameter is further apart than the four +-lr steps AdamW can take
    assert np.abs(pipe[0] - one[0]).max() <= 4 * 2 * 5e-4 * 1.1 + 1e-6          # (lr = 5e-4: at most +-lr per step on each side)


def test_replicas_stay_identical_when_ranks_start_apart_and_feed_data_before_the_first_step(tmp_path):
    """the capture warm-up used to apply two un-reduced optimizer steps on rank-local data and nothing
    broadcast rank 0's weights.  Now: broadcast at construction, warm-up restored in place."""
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    paths = [here, os.path.join(root, "iccv2025-upp_amd"), os.path.join(root, "oracle")]
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), paths, "rank_seeded"), nprocs=2, join=True)
    a = np.load(tmp_path / "params_rank_seeded_w2_r0.npy"); b = np.load(tmp_path / "params_rank_seeded_w2_r1.npy")
    np.testing.assert_array_equal(a, b)


def test_bench_gpus_2_reports_two_ranks(tmp_path):
    """`python bench.py --gpus 2` (no launcher around it) starts two ranks itself; gloo stands in for RCCL on a one-GPU box."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UPP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8",
                          ], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 16 and rec["dist_backend"] == "gloo" and rec["value"] > 0


def test_bench_runs_the_distributed_path_over_rccl_with_one_rank(tmp_path):
    """The one piece of the N > 1 path a one-GPU box cannot stand in for with gloo is RCCL itself: with UPP_FORCE_DIST=1 the bench (and
    TrainStep) take the distributed path with a ONE-rank "nccl" group -- RCCL initialises on the card, broadcasts the model, all-reduces
    the flat gradient buffer between the graph replays of every step, runs the barrier / max-over-ranks timing and the RCCL probe."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UPP_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 400))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "UPP_DIST_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8",
                          "--no-cpu-baseline", "--no-stage-report"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["dist_backend"] == "nccl" and rec["rccl_ranks"] == 1 and rec["value"] > 0
    # RCCL's version banner goes to stdout when the first communicator comes up: the line must still be the ONLY thing there
    assert [l for l in out.stdout.splitlines() if l.strip()] == [l for l in out.stdout.splitlines() if l.startswith("{")] and out.stdout.count("\n") == 1


def test_gradient_all_reduce_and_model_broadcast_run_over_a_one_rank_rccl_group(tmp_path):
    """... and the two collectives of the product's own step driver (utils.dist_utils.FlatGradAllReduce.reduce, train.broadcast_model) really
    issue under the rehearsal switch: counted through torch.distributed's wrappers in a child process with a one-rank nccl group."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import os, sys
sys.path[:0] = [%r, %r, %r]
import torch, torch.distributed as dist
dist.init_process_group(backend='nccl')
torch.cuda.set_device(0)
calls = {'all_reduce': 0, 'broadcast': 0}
real_ar, real_bc = dist.all_reduce, dist.broadcast
def ar(*a, **k):
    calls['all_reduce'] += 1
    return real_ar(*a, **k)
def bc(*a, **k):
    calls['broadcast'] += 1
    return real_bc(*a, **k)
dist.all_reduce, dist.broadcast = ar, bc
import _seeded
from models import build_model_from_cfg
from utils.config import builtin_cfg
from upp_hip.train import TrainStep, freeze_for_peft
m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda().train()
freeze_for_peft(m)
ts = TrainStep(m, (4, 1096, 3), use_graph=True)
assert ts.distributed
x = _seeded.noisy_clouds(4, 1024, seed=0).cuda(); y = torch.tensor([1, 2, 3, 4], device='cuda')
l0 = float(ts.step(x, y)); l1 = float(ts.step(x, y))
torch.cuda.synchronize()
assert calls['broadcast'] > 100 and calls['all_reduce'] == 2, calls
assert l0 == l0 and l1 == l1
print('OK', calls)
""" % (root, os.path.join(root, "iccv2025-upp_amd"), os.path.join(root, "tests"))
    env = dict(os.environ, UPP_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]


def test_learning_rate_changes_reach_a_captured_step():
    """lr / weight decay were by-value kernel arguments frozen into the optimizer graph."""
    import _seeded
    from models import build_model_from_cfg
    from upp_hip.train import TrainStep, freeze_for_peft
    from utils.config import builtin_cfg
    m = _seeded.fill(build_model_from_cfg(builtin_cfg('unify_modelnet_cls').model)).cuda().train()
    freeze_for_peft(m)
    ts = TrainStep(m, (4, 1096, 3), lr=1e-3)
    ts.pts.copy_(_seeded.noisy_clouds(4, 1024, seed=3).cuda())
    ts.labels.copy_(torch.tensor([1, 2, 3, 4], device='cuda'))
    ts.step()
    assert float(ts.opt.state[0]) == 1.0
    p0 = ts.opt.p.clone()
    ts.opt.set_lr(0.0, weight_decay=0.0)            # after capture
    ts.step()
    assert torch.equal(ts.opt.p, p0), "lr = 0 must freeze the parameters of a captured step"
    ts.opt.param_groups[-1]['lr'] = 1e-3
    ts.opt.param_groups[0]['lr'] = 1e-3
    ts.opt.sync_param_groups()
    ts.step()
    assert (ts.opt.p - p0).abs().max() > 0
    sd = ts.opt.state_dict()
    assert sd['param_groups'][0]['lr'] == 1e-3 and float(sd['state'][0]['step']) == 3.0
