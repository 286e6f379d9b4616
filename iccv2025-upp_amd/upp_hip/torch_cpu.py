"""Pure-torch grouping operators for CPU tensors -- BASELINE configs[0] ("Point_MAE_unify cls forward, 1 synthetic N = 1024 cloud on
torch-CPU: utils.misc.fps + torch.cdist kNN fallback -- plumbing, no GPU").  OFF by default: the product's operators have no CPU
path (`upp_hip.ops` raises on CPU tensors) and a HIP tensor never comes here.  `enable()` -- or UPP_TORCH_CPU=1 in the environment --
lets the four grouping entry points of upp_hip.functional (fps_gather, knn_query, knn_group, ChamferFunction) serve CPU tensors with the
torch formulations below, so that a reference user can run the model's forward on a GPU-less host to check plumbing (state-dict
loading, shapes, config wiring).  Nothing here touches oracle/ (test infrastructure) and nothing here is timed by bench.py.

Semantics follow the reference's own CPU-side formulations: FPS as datasets/ModelNetDataset.py:29-49 (start at index 0, arg-max of the
running minimum distance) with pointnet2_ops' rule that points with |p|^2 <= 1e-3 are never candidates; kNN as models/modules.py
knn_point (pairwise squared distances, k smallest, ties by index); Chamfer as a dense distance matrix with min / argmin.  Distances are
plain f32 torch arithmetic: on EXACT distance ties a pick can differ from the HIP kernels (which pin the CUDA kernels' fmaf order and
thread-strided tie rule: oracle/upp_oracle.c) -- bit-exact indices are a property of the HIP path, not of this fallback."""
import os

import torch

_ENABLED = os.environ.get("UPP_TORCH_CPU", "0") == "1"


def enable(on=True):
    global _ENABLED
    _ENABLED = bool(on)


def enabled():
    return _ENABLED


def fps(xyz, npoint):
    """xyz (B,N,3) f32 CPU -> (centres (B,npoint,3), idx (B,npoint) int32)."""
    B, N, _ = xyz.shape
    x = xyz.detach()
    live = (x * x).sum(-1) > 1e-3                        # pointnet2_ops: |p|^2 <= 1e-3 is skipped as a candidate
    dist = torch.full((B, N), 1e10, dtype=x.dtype)
    far = torch.zeros(B, dtype=torch.long)
    idx = torch.zeros(B, npoint, dtype=torch.long)
    rows = torch.arange(B)
    for j in range(npoint):
        idx[:, j] = far
        d = ((x - x[rows, far].unsqueeze(1)) ** 2).sum(-1)
        dist = torch.minimum(dist, d)
        far = torch.where(live, dist, torch.full_like(dist, -1.0)).argmax(-1)      # first maximum: lowest index
    centers = torch.gather(xyz, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
    return centers, idx.to(torch.int32)


def knn(ref, query, k):
    """ref (B,N,3), query (B,Q,3) -> (dist (B,Q,k) f32 EUCLIDEAN distances ascending, idx (B,Q,k) int64; ties by index) -- what KNN_CUDA
    (its sqrt kernel) and ops.knn (csrc/knn.hip sqrtf) return.  Ranked on the direct squared distance sum_c (q_c - r_c)^2, not on
    cdist ** 2 (whose rounding perturbs the tie order)."""
    q, r = query.detach(), ref.detach()
    d2 = ((q.unsqueeze(2) - r.unsqueeze(1)) ** 2).sum(-1)
    order = torch.argsort(d2, dim=-1, stable=True)[:, :, :k]
    return torch.gather(d2, -1, order).sqrt(), order


def knn_group(xyz, center, k):
    """-> (neighbourhood (B,G,k,3) centred on `center`, idx (B,G,k) int64); differentiable w.r.t. xyz and center."""
    _, idx = knn(xyz, center, k)
    B, G, _ = idx.shape
    nb = torch.gather(xyz.unsqueeze(1).expand(-1, G, -1, -1), 2, idx.unsqueeze(-1).expand(-1, -1, -1, 3))
    return nb - center.unsqueeze(2), idx


def chamfer(xyz1, xyz2):
    """-> (dist1 (B,N), dist2 (B,M)) squared nearest-neighbour distances, differentiable (reference extensions/chamfer_dist)."""
    d = ((xyz1.unsqueeze(2) - xyz2.unsqueeze(1)) ** 2).sum(-1)
    return d.min(2)[0], d.min(1)[0]
