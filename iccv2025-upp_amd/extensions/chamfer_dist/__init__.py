"""extensions.chamfer_dist (reference extensions/chamfer_dist/__init__.py:13-84)."""
import torch

from upp_hip.functional import ChamferFunction, chamfer_loss  # noqa: F401


def _strip_zero_points(xyz1, xyz2):
    # reference :37-41 -- only ever applied to a batch of one
    keep1 = torch.sum(xyz1, dim=2).ne(0)
    keep2 = torch.sum(xyz2, dim=2).ne(0)
    return xyz1[keep1].unsqueeze(dim=0), xyz2[keep2].unsqueeze(dim=0)


class _ChamferBase(torch.nn.Module):
    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def _fused(self, xyz1, xyz2, l1):
        """The module as one autograd node on the HIP path (upp_chamfer_loss), or None: the torch formulation follows."""
        if (xyz1.is_cuda and xyz2.is_cuda and xyz1.device == xyz2.device and xyz1.dtype == torch.float32 and xyz2.dtype == torch.float32
                and xyz1.dim() == 3 and xyz2.dim() == 3 and xyz1.shape[-1] == 3 and xyz2.shape[-1] == 3 and xyz1.size(0) == xyz2.size(0)
                and xyz1.size(0) > 0 and not (xyz1.size(0) == 1 and self.ignore_zeros)):
            return chamfer_loss(xyz1, xyz2, l1)
        return None

    def _dists(self, xyz1, xyz2):
        if xyz1.size(0) == 1 and self.ignore_zeros:
            xyz1, xyz2 = _strip_zero_points(xyz1, xyz2)
        if not xyz1.is_cuda:
            from upp_hip import torch_cpu           # opt-in torch formulation for CPU tensors (upp_hip.torch_cpu); ChamferFunction raises otherwise
            if torch_cpu.enabled():
                return torch_cpu.chamfer(xyz1, xyz2)
        return ChamferFunction.apply(xyz1, xyz2)


class ChamferDistanceL2(_ChamferBase):
    """mean(d1) + mean(d2) of squared nearest-neighbour distances."""

    def forward(self, xyz1, xyz2):
        fused = self._fused(xyz1, xyz2, False)
        if fused is not None:
            return fused
        d1, d2 = self._dists(xyz1, xyz2)
        return torch.mean(d1) + torch.mean(d2)


class ChamferDistanceL2_split(_ChamferBase):
    def forward(self, xyz1, xyz2):
        d1, d2 = self._dists(xyz1, xyz2)
        return torch.mean(d1), torch.mean(d2)


class ChamferDistanceL1(_ChamferBase):
    """(mean(sqrt d1) + mean(sqrt d2)) / 2."""

    def forward(self, xyz1, xyz2):
        fused = self._fused(xyz1, xyz2, True)
        if fused is not None:
            return fused
        d1, d2 = self._dists(xyz1, xyz2)
        return (torch.mean(torch.sqrt(d1)) + torch.mean(torch.sqrt(d2))) / 2
