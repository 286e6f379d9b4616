"""Evaluation loops of the reference classification runner (tools/runner_module.py:370-490), on device:
`validate` = one pass, arg-max accuracy; `test_vote` = the 10x voting protocol (FPS to a 1200-point superset once,
then `times` random 1024-subsets, each scale/translate-augmented; logits averaged before the arg-max)."""
import torch

from utils import dist_utils, misc
from upp_hip import functional as HF


def _accuracy(pred, label, distributed):
    pred, label = torch.cat(pred), torch.cat(label)
    if distributed:
        pred, label = dist_utils.gather_tensor(pred), dist_utils.gather_tensor(label)
    return (pred == label).sum() / float(label.size(0)) * 100.


@torch.no_grad()
def validate(model, batches, npoints, noisy=False, distributed=False):
    """batches: iterable of (points (B,N,3), label (B,)).  runner_module.py:383-413."""
    model.eval()
    preds, labels = [], []
    for points, label in batches:
        points = misc.fps(points.contiguous(), npoints)[0]
        logits = model(points, completion_prompt=noisy, denoise=noisy, point_num=npoints)
        preds.append(logits.argmax(-1).view(-1))
        labels.append(label.view(-1))
    return _accuracy(preds, labels, distributed)


@torch.no_grad()
def test_vote(model, batches, npoints, times=10, transform=misc.scale_translate, distributed=False, generator=None):
    """runner_module.py:427-490.  The random subsets are drawn on the device (torch.randperm) instead of
    np.random.choice on the host: same distribution, no host round trip per vote."""
    superset = {1024: 1200, 4096: 4800, 8192: 8192}
    if npoints not in superset:
        raise NotImplementedError()
    model.eval()
    preds, labels = [], []
    for points_raw, label in batches:
        point_all = min(superset[npoints], points_raw.size(1))
        raw, _ = HF.fps_gather(points_raw.contiguous(), point_all)                # (B, point_all, 3) FPS-ordered superset
        votes = []
        for _ in range(times):
            pick = torch.randperm(point_all, device=raw.device, generator=generator)[:npoints]
            points = raw[:, pick].contiguous()
            if transform is not None:
                points = transform(points)
            votes.append(model(points).unsqueeze(0))
        preds.append(torch.cat(votes, dim=0).mean(0).argmax(-1))
        labels.append(label.view(-1))
    return _accuracy(preds, labels, distributed)
