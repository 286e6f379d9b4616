"""Point_MAE_pretask_dev -- the prompter pre-training model (noise-vector supervision of the rectify prompter +
masked-shape completion through the pretask path and the MAE decoder), drop-in for reference
models/Point_MAE_pretask_dev.py:519-741: same registry name, constructor contract, forward kwargs and return values,
state-dict keys.  Its training recipe (reference tools/runner_pretask.py:157-247) is `pretask_losses` below: three
Chamfer-L1 terms on the gfx950 Chamfer kernels plus the noise loss returned by the model.
"""
import torch
import torch.nn as nn

from extensions.chamfer_dist import ChamferDistanceL1, ChamferDistanceL2
from utils import misc
from .build import MODELS
from . import upp_layers as L
from upp_hip import functional as HF
from .Point_MAE_unify import PromptedBackbone
from .upp_layers import Group


@MODELS.register_module()
class Point_MAE_pretask_dev(PromptedBackbone):
    def __init__(self, config):
        super().__init__()
        self._build_backbone(config, vis_base=config.num_group)

    def build_loss_func(self, loss_type):
        if loss_type == "cdl1":
            self.loss_func = ChamferDistanceL1()
        elif loss_type == 'cdl2':
            self.loss_func = ChamferDistanceL2()
        elif loss_type == 'emd':
            from emd import emd
            self.loss_func = emd()
        else:
            raise NotImplementedError

    def forward(self, pts, point_num=2048, train_with_gaussian=True, **kwargs):
        L.begin_forward(pts.device, self.training)
        try:
            return self._forward(pts, point_num=point_num, train_with_gaussian=train_with_gaussian, **kwargs)
        finally:
            L.end_forward()

    def _noise_supervision(self, pts, point_num, vis_center, tokens):
        """reference :669-702.  pts = [point_num shape points | noise points].  The rectify prompter predicts an offset per
        point; noise points are supervised with the mean offset to their 4 nearest shape points, shape points with 0."""
        P = pts.shape[1]
        pos = L.mlp2(self.pos_embed, vis_center)
        tokens = self.blocks(tokens, pos, path='rectify', rectify_adapter=True, rectify_prompts=True,
                             rectify_depth=self.config.prompter_config['rectify_depth'])
        noise, partial = pts[:, point_num:], pts[:, :point_num]
        pred = self.rectify_prompter(pts, vis_center, tokens, require_shape_feature=False)
        pred_pure, pred_noise = pred[:, :point_num], pred[:, point_num:]
        with torch.no_grad():
            # pytorch3d.ops.knn_points(noise, partial, K=4, return_nn=True) -> nn - noise, averaged over K:
            # exactly the centred neighbourhood the fused kNN+group kernel writes
            neigh, _ = L.OPS["knn_group"](partial.contiguous(), noise.contiguous(), 4)
            noise_vector = neigh.mean(dim=-2)
        if (pred.is_cuda and pred.dtype == torch.float32 and pred.shape[-1] == 3 and self.rectify_prompter.out_channels == 3
                and 0 < point_num < P and L.POOL_TRACE is None):
            # the two means of squared norms and the ranking score in two launches (+ one backward) instead of ~30 element-wise ones
            loss_pn, score = HF.noise_loss(pred, noise_vector, point_num)
            positive, negative = loss_pn, 0.0
        else:
            if self.rectify_prompter.out_channels == 1:
                positive = torch.mean((pred_noise - torch.norm(noise_vector, 2, dim=-1, keepdim=True)) ** 2)
            else:
                positive = torch.mean(torch.norm(pred_noise - noise_vector, 2, dim=-1, keepdim=True) ** 2)
            negative = torch.mean(torch.norm(pred_pure, 2, dim=-1, keepdim=True) ** 2)
            score = torch.norm(pred, p=2, dim=-1)
        order = HF.argsort_rows(score, descending=True)          # (rank-counting kernel: no library sort in the step)
        recall = torch.mean(torch.sum(order[:, :-point_num].detach() > point_num, dim=-1) / (P - point_num))
        kept = torch.gather(pts, 1, order[:, -point_num:, None].expand(-1, -1, 3)).detach()
        return positive + negative, recall, kept

    def _forward(self, pts, point_num=2048, train_with_gaussian=True, **kwargs):
        grouper = Group(num_group=self.vis_num, group_size=16)
        neighborhood, vis_center = grouper(pts)
        tokens = self.encoder(neighborhood)
        supervised = train_with_gaussian and self.training
        if supervised:
            noise_loss, recall, pts = self._noise_supervision(pts, point_num, vis_center, tokens)
            neighborhood, vis_center = grouper(pts)
            tokens = self.encoder(neighborhood)
        predict_center, rebuild_points = self._reconstruct(tokens, vis_center)
        if supervised:
            return predict_center, rebuild_points, noise_loss, recall
        return predict_center, rebuild_points


def pretask_losses(model, gt, partial, cropping, points, point_num, chamfer_l1=None):
    """One training step's loss of the pre-task recipe (reference tools/runner_pretask.py:210-225):
    `points` = partial cloud (+ appended noise points), `cropping` = the removed region, `gt` = the complete cloud.
    -> (total loss, dict of the individual terms)."""
    def cd_l1(a, b):                                   # ChamferDistanceL1 (extensions/chamfer_dist/__init__.py:61-73) on the op table
        from upp_hip import functional as HF
        if a.is_cuda and getattr(L.OPS["chamfer"], "__self__", None) is HF.ChamferFunction:        # (not replaced by a test's injection)
            return HF.chamfer_loss(a, b, True)           # (one node: upp_chamfer_fwd + upp_chamfer_loss, upp_chamfer_bwd)
        d1, d2 = L.OPS["chamfer"](a.contiguous(), b.contiguous())
        return (torch.mean(torch.sqrt(d1)) + torch.mean(torch.sqrt(d2))) / 2
    cd = chamfer_l1 if chamfer_l1 is not None else cd_l1
    out = model(points, point_num=point_num, train_with_gaussian=points.shape[1] > point_num, predict_center_num=16)
    if len(out) == 4:
        predict_center, rebuild, noise_loss, recall = out
    else:
        (predict_center, rebuild), noise_loss, recall = out, torch.zeros((), device=gt.device), torch.ones((), device=gt.device)
    coarse = cd(predict_center, cropping)
    crop_dense = cd(rebuild, cropping)
    dense = cd(torch.cat([partial, rebuild], dim=1), gt)
    total = coarse + crop_dense + dense + noise_loss
    return total, dict(cropping_coarse=coarse, cropping_dense=crop_dense, dense=dense, noise=noise_loss, recall=recall)
