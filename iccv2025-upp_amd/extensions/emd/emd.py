"""extensions.emd.emd (reference extensions/emd/emd.py:5-49)."""
import torch

from upp_hip.functional import EarthMoverDistanceFunction  # noqa: F401


class earth_mover_distance(torch.nn.Module):
    """forward(xyz1 (b,n1,3), xyz2 (b,n2,3), transpose=False) -> scalar mean of cost / n1."""

    def forward(self, xyz1, xyz2, transpose=False):
        cost = EarthMoverDistanceFunction.apply(xyz1, xyz2)
        return (cost / xyz1.size(1)).mean()
