"""Hot-path helpers of the reference's utils/misc.py."""
import torch

from upp_hip import functional as _F


def fps(data, number):
    """data (B,N,3), number -> (fps_data (B,number,3), fps_idx (B,number) int32).

    Reference utils/misc.py:13-20 runs furthest_point_sample, then gather_operation on a
    transposed copy and transposes back; here the sampled coordinates come out of the FPS
    kernel itself (same values: a gather is an exact copy)."""
    fps_data, fps_idx = _F.fps_gather(data.contiguous(), number)
    return fps_data, fps_idx


def peft_detect(name, targets):
    """reference utils/misc.py:22-26"""
    return any(t in name for t in targets)


def lidar_noise(points, number=64, scale=1.3, low=1.02, generator=None):
    """Outlier points p * U(low, scale) of `number` random existing points (reference
    utils/misc.py:38-46; host numpy RNG there, device RNG here)."""
    B, P, _ = points.shape
    idx = torch.randint(0, P, (number,), device=points.device, generator=generator)
    factor = torch.empty((1, number, 1), device=points.device).uniform_(low, scale, generator=generator)
    return points[:, idx, :] * factor


def gaussian_noise(shape, loc=0., scale=0.2, shell_radius=0.9, device=None, generator=None):
    """Shell noise g + shell_radius * g/|g|, g ~ N(loc, scale) (reference utils/misc.py:28-36)."""
    g = torch.empty(shape, device=device).normal_(loc, scale, generator=generator)
    return g + g / g.norm(p=2, dim=-1, keepdim=True) * shell_radius
