"""Point_MAE_unify_seg -- UPP part segmentation (reference models/Point_MAE_unify_segment.py:329-625,
cfgs/unify_shapenetpart_seg.yaml): the prompted backbone of Point_MAE_unify on 128 groups, block outputs 3 / 7 / 11
concatenated (1152-d), 3-NN feature propagation to the label points and a per-point head -> log-probabilities.

Same registry name, constructor, forward signature and state-dict keys as the reference.  The per-point head is
evaluated as row GEMMs; the 3456-wide input of its first layer is [per-point 1024 | per-sample 2432], so the
per-sample part is multiplied once per sample and broadcast (3.4x fewer FLOPs than the concat the reference builds).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .build import MODELS
from . import upp_layers as L
from upp_hip import functional as HF
from .Point_MAE_unify import PromptedBackbone
from .upp_layers import Group, PointNetFeaturePropagation, PositionalEmbedding, _bn_rows, _pointwise_bn_relu


class get_loss(nn.Module):
    def forward(self, pred, target):
        if pred.is_cuda and pred.dim() == 2:
            # F.nll_loss reduces 65,536 rows in a single workgroup on this stack (0.1 ms); here two launches with fixed-order sums
            # forward and one dense write backward (upp_nll_mean_fwd / _bwd) -- torch: gather, mean, neg, div, zero-fill + scatter_add
            return HF.nll_mean(pred, target)
        return F.nll_loss(pred, target)


@MODELS.register_module()
class Point_MAE_unify_seg(PromptedBackbone):
    def __init__(self, config):
        super().__init__()
        D = self._build_backbone(config)
        self.label_conv = nn.Sequential(nn.Conv1d(16, 64, kernel_size=1, bias=True), nn.BatchNorm1d(64), nn.LeakyReLU(0.2),
                                        nn.Conv1d(64, 128, kernel_size=1, bias=True), nn.BatchNorm1d(128), nn.LeakyReLU(0.2))
        self.positional_embedding = PositionalEmbedding(12)
        self.propagation_0 = PointNetFeaturePropagation(in_channel=D * 3 + 3, mlp=[D * 4, 1024], interpolate_neighbors=3)
        self.group_divider1 = Group(num_group=self.num_group * 4, group_size=self.group_size // 2)
        self.seg_head = nn.Sequential(
            nn.Conv1d(1024 + 128 + D * 6, 512, 1), nn.BatchNorm1d(512), nn.ReLU(), nn.Dropout(0.5),
            nn.Conv1d(512, 256, 1), nn.BatchNorm1d(256), nn.ReLU(),
            nn.Conv1d(256, self.cls_dim, 1))
        self.get_loss = get_loss()

    def _label_feature(self, cls_label):
        """label_conv on the (B,16) one-hot object class -> (B,128); BatchNorm over the batch."""
        c1, b1, a1, c2, b2, a2 = self.label_conv
        x = cls_label.reshape(cls_label.shape[0], 16).to(self.label_conv[0].weight.dtype)      # (.float() in the reference; f64 in the tests' arbitration runs)
        for conv, bn, act in ((c1, b1, a1), (c2, b2, a2)):
            if self.training and bn.track_running_stats:
                L.bump_counter(bn.num_batches_tracked)
            z = _bn_rows(HF.linear(x, conv.weight.squeeze(-1), conv.bias, own_wgrad=True), bn, self.training)
            x = L.gate('label_conv', z, act.negative_slope) if (L.POOL_TRACE is not None and isinstance(act, nn.LeakyReLU)) else act(z)
        return x

    def _head(self, point_feat, global_feat):
        """seg_head on cat([point_feat (B,N,1024), global_feat (B,2432) broadcast]) -> (B,N,cls) log-probabilities."""
        c1, bn1, _, drop, c2, bn2, _, c3 = self.seg_head
        B, N, C = point_feat.shape
        w1 = c1.weight.squeeze(-1)
        per_sample = HF.linear(global_feat, w1[:, C:], c1.bias, own_wgrad=True)                       # (B,512), once per sample
        h = HF.linear_group_bias(point_feat.reshape(B * N, C), w1[:, :C], per_sample, N).view(B, N, -1) if (N & (N - 1)) == 0 and N >= 32 \
            else HF.linear(point_feat.reshape(B * N, C), w1[:, :C]).view(B, N, -1) + per_sample.unsqueeze(1)    # (per-sample term: GEMM epilogue)
        if self.training and bn1.track_running_stats:
            L.bump_counter(bn1.num_batches_tracked)
        h = _bn_rows(h.view(B * N, -1), bn1, self.training, relu=True, drop=drop)      # (the dropout rides in the BatchNorm's passes)
        h = _pointwise_bn_relu(h, c2, bn2, self.training)
        w3 = c3.weight.squeeze(-1)
        if L.POOL_TRACE is None and HF.linear_pad_n_usable(h, w3):
            # the 50 part classes: the GEMM writes a 52-column matrix from the un-padded weight's plane image; bias + log-softmax read it
            # where it lies (no padded weight / bias copies, no [:, :50] slice and its backward)
            ypad = HF._LinearPadN.apply(h, w3) if (torch.is_grad_enabled() and (h.requires_grad or w3.requires_grad)) else \
                HF.ops.linear_f32(h, None, None, HF.ops.LIN_NONE, planes=HF.ops.PLANES.get(w3) if not w3.requires_grad else HF.ops.PLANES._split(w3.detach()),
                                  wshape=((w3.shape[0] + 3) // 4 * 4, w3.shape[1]))
            return HF.log_softmax_rows(ypad, c3.bias, w3.shape[0]).view(B, N, -1)
        h = HF.linear(h, w3, c3.bias)
        return F.log_softmax(h, dim=-1).view(B, N, -1)

    def forward(self, pts, cls_label, label_points=None, completion_prompt=True, denoise=True, point_num=1024, **kwargs):
        L.begin_forward(pts.device, self.training)
        try:
            return self._forward(pts, cls_label, label_points=label_points, completion_prompt=completion_prompt, denoise=denoise, point_num=point_num, **kwargs)
        finally:
            L.end_forward()

    def _forward(self, pts, cls_label, label_points=None, completion_prompt=True, denoise=True, point_num=1024, **kwargs):
        return self._back(self._front(pts, completion_prompt, denoise, point_num), cls_label, label_points)

    # -- front-end / back-end split (upp_hip.train.PipelinedTrainStep): the front-end reads no PEFT-trainable parameter ----
    def _front(self, pts, completion_prompt, denoise, point_num):
        """Prompting front-end, grouping and patch embedding of the prompted cloud, positional embedding and the level-2
        grouping -> (prompted pts, tokens, centres, pos[, centre2, centre1_idx, centre2_idx])."""
        if denoise:
            pts = self._rectify(pts, point_num)
            if L.DIAG_AUX:
                self.aux['dbg_rectified'] = pts
        if completion_prompt:
            pts = self._complete(pts, point_num)
        neighborhood, center = self.group_divider(pts)
        state = (pts, self.encoder(neighborhood), center, L.mlp2(self.pos_embed, center))
        lvl2 = self._level2(center)
        if lvl2:
            state += (lvl2['center2'], lvl2['center1_idx'], lvl2['center2_idx'])
        return state

    def _back(self, state, cls_label, label_points=None):
        pts, tokens, center, pos = state[:4]
        propagation = {}
        if len(state) >= 7:
            propagation = dict(center1=center, center1_idx=state[5], center2=state[4], center2_idx=state[6],
                               gather_idx=self.config.gather_idx, prompt_propagation_after=self.config.prompt_propagation_after)
        pc = self.config.prompter_config
        feats = self.blocks(tokens, pos, path='downstream', downstream_adapter=pc.downstream_adapter,
                            downstream_prompts=pc.downstream_prompts, classification=False, feature_list=True, **propagation)
        x = torch.cat(feats, dim=-1)                                                  # (B,G,1152)
        global_feat = torch.cat((L.max_over(x, 1, 'seg.global_max'), torch.mean(x, 1), self._label_feature(cls_label)), -1)   # (B,2432)
        target = label_points if label_points is not None else pts
        f0 = self.propagation_0(target, center, target, x)                            # (B,N,1024)
        return self._head(f0, global_feat)

    def prompt_tokens(self, pts, completion_prompt=True, denoise=True, point_num=1024):
        L.begin_forward(pts.device, self.training)
        try:
            return self._front(pts, completion_prompt, denoise, point_num)
        finally:
            L.end_forward()

    def forward_tokens(self, state, cls_label, label_points=None):
        L.begin_forward(state[1].device, self.training)
        try:
            return self._back(tuple(state), cls_label, label_points)
        finally:
            L.end_forward()
