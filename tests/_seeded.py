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


# The cloud generators live with the product (iccv2025-upp_amd/utils/synthetic.py: bench.py uses them).  Loaded by file path: the
# fixture generators under oracle/ import this module next to the REFERENCE's own `utils` package.
import importlib.util as _ilu  # noqa: E402
import os as _os  # noqa: E402

_spec = _ilu.spec_from_file_location("upp_synthetic", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                                    "iccv2025-upp_amd", "utils", "synthetic.py"))
_syn = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(_syn)
noisy_clouds, unit_ball_clouds = _syn.noisy_clouds, _syn.unit_ball_clouds


def ref_ops_cases(fixture):
    """(name, clouds (B,N,3) f32 tensor, M, Q, k) for every case of tests/golden/ref_ops.npz (oracle/gen_golden_ops.py)."""
    for line in fixture["cases"]:
        name, seed, B, N, M, Q, k = str(line).split(",")
        seed, B, N, M, Q, k = int(seed), int(B), int(N), int(M), int(Q), int(k)
        pts = noisy_clouds(B, 1024, seed=seed) if N == 1096 else unit_ball_clouds(B, N, seed=seed)
        yield name, pts, M, Q, k
