import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "iccv2025-upp_amd")
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture
def oracle_ops(monkeypatch):
    """Run the model's grouping on the CPU oracle (GPU-less host).  Test injection only."""
    import oracle
    from models import upp_layers
    from upp_hip import functional as HF
    ops = oracle.torch_ops()
    monkeypatch.setitem(upp_layers.OPS, "fps_gather", ops["fps_gather"])
    monkeypatch.setitem(upp_layers.OPS, "knn_group", ops["knn_group"])
    monkeypatch.setitem(upp_layers.OPS, "chamfer", ops["chamfer"])
    monkeypatch.setattr(HF, "fps_gather", ops["fps_gather"])  # utils.misc.fps resolves through HF
    return ops


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return {n: np.load(os.path.join(GOLDEN, n + ".npz")) for n in ("upp_model", "upp_modules", "upp_seg", "point_mae", "pretask", "ref_ops", "upp_stage2", "upp_seg_train", "upp_stage2_f64", "upp_seg_train_f64")}
