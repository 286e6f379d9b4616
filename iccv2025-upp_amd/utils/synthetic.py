"""Synthetic inputs of SURVEY 8(d) -- seeded point clouds for the benchmark, the profiling tools and the tests (there are no
datasets in the build or on the GPU box).  Reference recipes: datasets/ModelNetDataset.py:20-25 (pc_normalize),
tools/runner_module.py:160-169 and utils/misc.py:28-46 (noisy-train input)."""
import torch


def unit_ball_clouds(B, N, seed=0):
    """Synthetic clouds of SURVEY 8(d): uniform in the unit ball, then centred and scaled to
    max-norm 1 (ModelNet pc_normalize semantics)."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(B, N, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    r = torch.rand(B, N, 1, generator=g) ** (1.0 / 3.0)
    p = d * r
    p = p - p.mean(dim=1, keepdim=True)
    p = p / p.norm(dim=-1).max(dim=1)[0].view(B, 1, 1)
    return p.contiguous()


def noisy_clouds(B, N=1024, seed=0, lidar=48, gauss=24):
    """Noisy-train input of tools/runner_module.py:160-169: N clean + 48 'lidar' outliers
    (p * U(1.2,1.5)) + 24 shell-Gaussian points -> (B, N+72, 3)."""
    g = torch.Generator().manual_seed(seed + 1000)
    p = unit_ball_clouds(B, N, seed)
    idx = torch.randint(0, N, (lidar,), generator=g)
    fac = torch.empty(1, lidar, 1).uniform_(1.2, 1.5, generator=g)
    lid = p[:, idx, :] * fac
    gn = torch.empty(B, gauss, 3).normal_(0., 0.1, generator=g)
    gn = gn + gn / gn.norm(dim=-1, keepdim=True) * 0.9
    return torch.cat([p, lid, gn], dim=1).contiguous()
