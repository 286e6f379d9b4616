"""Hot-path helpers of the reference's utils/misc.py."""
import torch

from upp_hip import functional as _F


def fps(data, number):
    """data (B,N,3), number -> (fps_data (B,number,3), fps_idx (B,number) int32).

    Reference utils/misc.py:13-20 runs furthest_point_sample, then gather_operation on a
    transposed copy and transposes back; here the sampled coordinates come out of the FPS
    kernel itself (same values: a gather is an exact copy)."""
    fps_data, fps_idx = _F.fps_gather(data.contiguous(), number)
    from models import upp_layers as L
    if L.POOL_TRACE is not None:          # (test instrument, see upp_layers.trace_idx)
        fps_idx = L.trace_idx('misc.fps', fps_idx)
        fps_data = L.index_points(data, fps_idx.long())
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


def seprate_point_cloud(xyz, num_points, crop, fixed_points=None, padding_zeros=False, sample_points=1024,
                        incomplete_shape=True, centers=None, generator=None):
    """Online cropping of reference utils/misc.py:205-256, batched on the device.

    The reference loops over the batch in Python (randn centre -> distance argsort -> slice -> one single-cloud FPS
    launch for the kept part and one for the cropped part, 2*B launches of up to 1023 rounds each).  All samples have
    the same sizes, so here the whole batch is ONE argsort, two gathers and two batched FPS launches.
    Returns (input_data (B, min(n-crop, sample_points), 3), crop_data (B, min(crop, sample_points), 3)).
    `centers` (B,1,3) overrides the random viewpoints (the reference's `fixed_points`, one per sample)."""
    B, n, c = xyz.shape
    assert n == num_points and c == 3
    if crop == num_points:
        return xyz, None
    if isinstance(crop, (list, tuple)):
        num_crop = int(torch.randint(crop[0], crop[1] + 1, (1,), generator=generator).item())
    else:
        num_crop = crop
    if centers is None:
        if fixed_points is not None:
            fp = fixed_points[torch.randint(0, len(fixed_points), (1,)).item()] if isinstance(fixed_points, list) else fixed_points
            centers = fp.reshape(1, 1, 3).to(xyz.device).expand(B, 1, 3)
        else:
            centers = torch.nn.functional.normalize(torch.randn(B, 1, 3, device=xyz.device, generator=generator), p=2, dim=-1)
    dist = torch.norm(centers - xyz, p=2, dim=-1)                                   # (B, n)
    order = _F.argsort_rows(dist)                     # (rank-counting kernel on the device, torch.argsort on the host)
    take = lambda idx: torch.gather(xyz, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    crop_data = take(order[:, :num_crop])
    if padding_zeros:
        input_data = xyz.clone()
        input_data.scatter_(1, order[:, :num_crop].unsqueeze(-1).expand(-1, -1, 3), 0.0)
    else:
        input_data = take(order[:, num_crop:])
    force = isinstance(crop, (list, tuple))
    if (incomplete_shape or force) and input_data.shape[1] > sample_points:
        input_data = fps(input_data, sample_points)[0]
    if (incomplete_shape or force) and crop_data.shape[1] > sample_points:
        crop_data = fps(crop_data, sample_points)[0]
    return input_data.contiguous(), crop_data.contiguous()


def scale_translate(pc, scale_low=2. / 3., scale_high=3. / 2., translate_range=0.2, generator=None):
    """PointcloudScaleAndTranslate (reference datasets/data_transforms.py:54-68) without the per-sample Python loop and
    host RNG: one anisotropic scale in [2/3, 3/2]^3 and one translation in [-0.2, 0.2]^3 per cloud, on the device."""
    B = pc.shape[0]
    s = torch.empty(B, 1, 3, device=pc.device).uniform_(scale_low, scale_high, generator=generator)
    t = torch.empty(B, 1, 3, device=pc.device).uniform_(-translate_range, translate_range, generator=generator)
    return pc * s + t


def noisy_train_batch(points, npoints=1024, crop_ratio=0.25, lidar=48, gauss=24, generator=None):
    """The per-step input recipe of reference tools/runner_module.py:126-186 for `noisy_train` + `incomplete_cropping`:
    (B, N_full, 3) clean clouds -> cropped + FPS-resampled (B, npoints, 3) -> + lidar outliers + shell noise ->
    scale/translate.  Everything stays on the device."""
    partial, _ = seprate_point_cloud(points, points.shape[1], int(points.shape[1] * crop_ratio), sample_points=npoints,
                                     generator=generator)
    out = [partial]
    if lidar:
        out.append(lidar_noise(partial, lidar, low=1.2, scale=1.5, generator=generator))
    if gauss:
        out.append(gaussian_noise([partial.shape[0], gauss, 3], loc=0., scale=0.1, shell_radius=0.9, device=partial.device,
                                  generator=generator))
    return scale_translate(torch.cat(out, dim=1), generator=generator).contiguous()
