"""Drop-in for KNN_CUDA 0.2 (`from knn_cuda import KNN`, reference models/Point_MAE_unify.py:16,56,69).

Unlike upstream it imports on a host without a GPU (upstream asserts CUDA and JIT-compiles
at import time); calling it with CPU tensors raises unless the opt-in torch formulation is on (upp_hip.torch_cpu.enable() /
UPP_TORCH_CPU=1: torch.cdist + stable argsort, BASELINE configs[0])."""
import torch
import torch.nn as nn

from upp_hip import ops

__version__ = "0.2+upp_hip"


def _t(x):
    return x.transpose(-1, -2).contiguous()


class KNN(nn.Module):
    """forward(ref, query) -> (D f32, I int64), neighbours ordered by (distance, index).

    transpose_mode=True : ref (B,N,dim), query (B,Q,dim) -> (B,Q,k) each.
    transpose_mode=False: ref (B,dim,N), query (B,dim,Q) -> (B,k,Q) each.
    dim must be 3 (the only case UPP uses)."""

    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k = k
        self._t = transpose_mode

    def forward(self, ref, query):
        assert ref.size(0) == query.size(0), "ref.shape={} != query.shape={}".format(ref.shape, query.shape)
        with torch.no_grad():
            if not self._t:
                ref, query = _t(ref), _t(query)
            if not ref.is_cuda:
                from upp_hip import torch_cpu       # opt-in torch formulation for CPU tensors (BASELINE configs[0]); ops raises otherwise
                if torch_cpu.enabled():
                    d, i = torch_cpu.knn(ref.float(), query.float(), self.k)
                    return (d, i) if self._t else (_t(d), _t(i))
            d, i, _ = ops.knn(ref.float().contiguous(), query.float().contiguous(), self.k)
            if not self._t:
                d, i = _t(d), _t(i)
        return d, i
