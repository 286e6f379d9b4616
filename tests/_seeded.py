"""Key-seeded weights shared by the golden generator (reference model, build container) and
the parity tests (this repo's model, anywhere): every state-dict entry is filled from a
generator seeded by crc32(key), so the values do not depend on module construction order
and no 122 MB checkpoint has to be committed."""
import zlib

import torch


def seeded_state_dict(model):
    out = {}
    for key, ref in model.state_dict().items():
        g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
        shape = tuple(ref.shape)
        if key.endswith('num_batches_tracked'):
            t = torch.zeros(shape, dtype=ref.dtype)
        elif key.endswith('running_var'):
            t = 1 + 0.1 * torch.rand(shape, generator=g)
        elif key.endswith('running_mean'):
            t = 0.1 * torch.randn(shape, generator=g)
        elif ref.dim() >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) / max(fan_in, 1) ** 0.5
        elif key.endswith('bias'):
            t = 0.1 * torch.randn(shape, generator=g)
        else:  # 1-D scale of a normalisation layer
            t = 1 + 0.1 * torch.randn(shape, generator=g)
        out[key] = t.to(ref.dtype)
    return out


def fill(model):
    model.load_state_dict(seeded_state_dict(model), strict=True)
    return model


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


def ref_ops_cases(fixture):
    """(name, clouds (B,N,3) f32 tensor, M, Q, k) for every case of tests/golden/ref_ops.npz (oracle/gen_golden_ops.py)."""
    for line in fixture["cases"]:
        name, seed, B, N, M, Q, k = str(line).split(",")
        seed, B, N, M, Q, k = int(seed), int(B), int(N), int(M), int(Q), int(k)
        pts = noisy_clouds(B, 1024, seed=seed) if N == 1096 else unit_ball_clouds(B, N, seed=seed)
        yield name, pts, M, Q, k
