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


def _worker(rank, world, port, out_dir, paths):
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
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, 'drop_prob'):
            mod.drop_prob = 0.0
    freeze_for_peft(m)
    ts = PipelinedTrainStep(m, (4, 1096, 3))
    assert ts.distributed == (world > 1)
    for k in range(4):
        pts = _seeded.noisy_clouds(4, 1024, seed=100 * rank + k).cuda()
        labels = torch.tensor([(rank + k) % 40, 3, 17, 39 - k], device='cuda')
        ts.step(pts, labels)
    ts.flush()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in ts.trainable]).cpu().numpy()
    np.save(os.path.join(out_dir, "params_w%d_r%d.npy" % (world, rank)), flat)
    np.save(os.path.join(out_dir, "loss_w%d_r%d.npy" % (world, rank)), np.array(float(ts.loss)))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def test_two_ranks_keep_identical_parameters(tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    paths = [here, os.path.join(root, "iccv2025-upp_amd"), os.path.join(root, "oracle")]
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), paths), nprocs=2, join=True)
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), paths), nprocs=1, join=True)
    a = np.load(tmp_path / "params_w2_r0.npy"); b = np.load(tmp_path / "params_w2_r1.npy"); solo = np.load(tmp_path / "params_w1_r0.npy")
    np.testing.assert_array_equal(a, b)                       # one all-reduce per step: bit-identical replicas
    assert np.abs(a - solo).max() > 1e-6                      # ... and the other rank's batches did contribute
    assert np.isfinite(np.load(tmp_path / "loss_w2_r0.npy")) and np.isfinite(np.load(tmp_path / "loss_w2_r1.npy"))
