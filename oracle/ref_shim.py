"""Import the reference's own model code (read-only /root/reference) in THIS container.

TEST INFRASTRUCTURE, build-container only: used by oracle/gen_golden.py to produce the
fixtures under tests/golden/.  Nothing here travels to the GPU box as an import: tests read
only the .npz fixtures.

The reference cannot be imported as shipped (circular import between
Point_MAE_unify and Point_MAE_pretask_dev, undefined `pooling`, missing third-party CUDA
packages).  The shim:
  * stubs timm / easydict / ipdb / termcolor / pytorch3d / chamfer / emd_cuda,
  * serves pointnet2_ops and knn_cuda from the CPU oracle (oracle/upp_oracle.c),
  * breaks the import cycle in two phases and supplies `pooling` (SURVEY D.3 assumption).
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REF, "models"))


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = EasyDict(v) if isinstance(v, dict) else v
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class DropPath(nn.Module):  # timm 0.4.5 semantics
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.drop_prob or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = (keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)).floor_()
        return x.div(keep) * mask


def pooling(x, transform):
    """SURVEY D.3 (assumption; `pooling` is undefined in the reference)."""
    lc = x.max(dim=2)[0] + x.mean(dim=2)
    return transform(lc.permute(0, 2, 1)).permute(0, 2, 1)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_loaded = None


def load():
    """-> namespace with .uni (models.Point_MAE_unify), .dev (models.Point_MAE_pretask_dev), .MODELS, .EasyDict"""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present")
    sys.path.insert(0, _HERE)
    import oracle as O

    class KNN(nn.Module):
        def __init__(self, k, transpose_mode=False):
            super().__init__()
            self.k, self.t = k, transpose_mode

        def forward(self, ref, query):
            assert self.t
            d, i = O.knn(ref.detach().numpy(), query.detach().numpy(), self.k)
            return torch.from_numpy(d).to(ref.dtype), torch.from_numpy(i)        # (f64 arbitration runs: indices from the f32 image)

    def furthest_point_sample(xyz, npoint):
        return torch.from_numpy(O.fps(xyz.detach().numpy(), int(npoint)))

    def gather_operation(features, idx):
        return torch.gather(features, 2, idx.long().unsqueeze(1).expand(-1, features.shape[1], -1))

    _mod('timm'); _mod('timm.models')
    _mod('timm.models.layers', DropPath=DropPath, trunc_normal_=lambda t, std=.02: nn.init.trunc_normal_(t, std=std))
    _mod('timm.scheduler', CosineLRScheduler=object)
    _mod('ipdb'); _mod('termcolor', colored=lambda s, *a, **k: s)
    _mod('easydict', EasyDict=EasyDict)
    def _ch_fwd(a, b):
        return [torch.from_numpy(t) for t in O.chamfer_fwd(a.detach().numpy(), b.detach().numpy())]

    def _ch_bwd(a, b, i1, i2, g1, g2):
        return [torch.from_numpy(t) for t in O.chamfer_bwd(a.detach().numpy(), b.detach().numpy(), i1.numpy(), i2.numpy(),
                                                           g1.contiguous().numpy(), g2.contiguous().numpy())]

    def knn_points(p1, p2, K=1, return_nn=False, **_):
        """pytorch3d.ops.knn_points restated on the oracle kNN: K nearest p2 points of every p1 point, ascending distance."""
        d, i = O.knn(p2.detach().numpy(), p1.detach().numpy(), K)            # (B, n1, K) distances (sqrt) and indices
        i = torch.from_numpy(i)
        nn_pts = torch.gather(p2.unsqueeze(1).expand(-1, p1.shape[1], -1, -1), 2, i.unsqueeze(-1).expand(-1, -1, -1, 3)) if return_nn else None
        return torch.from_numpy(d).to(p1.dtype) ** 2, i, nn_pts

    p3 = _mod('pytorch3d.ops', knn_points=knn_points)
    _mod('pytorch3d', ops=p3); _mod('chamfer', forward=_ch_fwd, backward=_ch_bwd); _mod('emd_cuda')
    _mod('knn_cuda', KNN=KNN)
    p2u = _mod('pointnet2_ops.pointnet2_utils', furthest_point_sample=furthest_point_sample,
               gather_operation=gather_operation)
    _mod('pointnet2_ops', pointnet2_utils=p2u)
    # keep our own drop-in packages out of the way: the reference must resolve ITS utils/models/extensions
    for name in list(sys.modules):
        if name.split('.')[0] in ('utils', 'models', 'extensions', 'emd'):
            del sys.modules[name]
    sys.path[:] = [p for p in sys.path if 'iccv2025-upp_amd' not in p]
    sys.path[:0] = [REF, os.path.join(REF, 'extensions')]
    pkg = types.ModuleType('models'); pkg.__path__ = [os.path.join(REF, 'models')]
    sys.modules['models'] = pkg
    fake = types.ModuleType('models.Point_MAE_unify')
    fake.Group = fake.propagate = fake.pooling = None
    sys.modules['models.Point_MAE_unify'] = fake
    dev = importlib.import_module('models.Point_MAE_pretask_dev')
    del sys.modules['models.Point_MAE_unify']
    uni = importlib.import_module('models.Point_MAE_unify')
    dev.Group, dev.propagate, dev.pooling = uni.Group, uni.propagate, pooling
    torch.cuda.empty_cache = lambda: None          # called unconditionally at Point_MAE_unify_segment.py:590
    seg = importlib.import_module('models.Point_MAE_unify_segment')
    nn.Module.cuda = lambda self, *a, **k: self     # Point_MAE_cp.py:409 moves its (parameter-less) loss module to cuda
    cp = importlib.import_module('models.Point_MAE_cp')
    from models.build import MODELS
    _loaded = types.SimpleNamespace(uni=uni, dev=dev, seg=seg, cp=cp, MODELS=MODELS, EasyDict=EasyDict, pooling=pooling)
    return _loaded


def model_cfg(name='unify_modelnet_cls'):
    import yaml
    with open(os.path.join(REF, 'cfgs', name + '.yaml')) as f:
        return EasyDict(yaml.safe_load(f)['model'])
