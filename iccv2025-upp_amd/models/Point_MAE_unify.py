"""Point_MAE_unify -- the UPP classifier (rectify prompter -> completion prompter -> downstream
Point-MAE backbone with prompts/adapters), drop-in for reference models/Point_MAE_unify.py:390-655:
same registry name, constructor contract `cls(config)`, forward kwargs, methods and state-dict
keys; grouping runs on the gfx950 FPS / kNN kernels of libupp_hip.so.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from utils import misc
from .build import MODELS
from . import upp_layers as L
from upp_hip import functional as HF
from .upp_layers import (  # noqa: F401  (re-exported: the reference's sibling modules import these from here)
    Block, Encoder, Group, RectifyPrompter, TransformerDecoder, TransformerEncoder,
    pooling, propagate, trunc_normal_,
)


def _mlp2(i, h, o):
    return nn.Sequential(nn.Linear(i, h), nn.GELU(), nn.Linear(h, o))


class PromptedBackbone(nn.Module):
    """Everything Point_MAE_unify and Point_MAE_unify_seg share: patch embedding, the 12-block prompted
    transformer, the rectify (denoising) prompter and the completion prompter with its MAE decoder
    (reference models/Point_MAE_unify.py:392-474,541-610 == models/Point_MAE_unify_segment.py:330-408,482-561)."""

    def _build_backbone(self, config, vis_base=64):
        """vis_base: the group count the visible/masked split is derived from -- hard-wired 64 in the unify models
        (reference Point_MAE_unify.py:404), `num_group` in the pre-task model (Point_MAE_pretask_dev.py:533)."""
        self.config = config
        self.aux = {}     # by-products of the last forward (not state): 'rebuild_points' of the completion prompter
        tc = config.transformer_config
        self.trans_dim = tc.trans_dim
        self.mask_ratio = tc.mask_ratio
        self.depth = tc.depth
        self.num_heads = tc.num_heads
        self.encoder_dims = tc.encoder_dims
        self.drop_path_rate = tc.drop_path_rate
        self.group_size = config.group_size
        self.num_group = config.num_group
        self.vis_num = vis_base - int(self.mask_ratio * vis_base)
        self.n_masked = int(vis_base - self.vis_num)
        self.vis_short = 16
        self.cls_dim = config.get('cls_dim') if hasattr(config, 'get') else getattr(config, 'cls_dim', None)
        D = self.trans_dim

        self.encoder = Encoder(encoder_channel=self.encoder_dims)
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        self.blocks = TransformerEncoder(embed_dim=D, depth=self.depth, drop_path_rate=dpr,
                                         num_heads=self.num_heads, **self.config.prompter_config)
        self.norm = nn.LayerNorm(D)
        self.shape_pred = _mlp2(D, D // 2, self.vis_short)
        self.coarse_pred = _mlp2(self.vis_short * self.vis_num, D, 3 * self.n_masked)
        self.predict_token_generator = _mlp2(D, 128, D)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, D))
        self.decoder_pos_embed = _mlp2(3, 128, D)
        self.decoder_depth = tc.decoder_depth
        self.decoder_num_heads = tc.decoder_num_heads
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.decoder_depth)]
        self.MAE_decoder = TransformerDecoder(embed_dim=D, depth=self.decoder_depth, drop_path_rate=dpr,
                                              num_heads=self.decoder_num_heads, pretask_adapter=True, pretask_depth=4)
        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size)
        self.dense_pred = nn.Sequential(nn.Conv1d(D, 3 * self.group_size, 1))
        self.rectify_prompter = RectifyPrompter(in_channels=3, out_channels=3, hidden_dimesion=D, embedding_level=4,
                                                num_group=32, group_size=16, top_center_dim=12)
        self.pos_embed = _mlp2(3, 128, D)
        trunc_normal_(self.mask_token, std=.02)
        return D

    def load_model_from_ckpt(self, bert_ckpt_path, logger=None):
        """Key rewrites of reference :505-536: strip 'module.', '_block', 'MAE_encoder.', 'base_model.'."""
        if bert_ckpt_path is None:
            return None
        ckpt = torch.load(bert_ckpt_path, map_location='cpu')
        base = {k.replace("module.", "").replace('_block', ''): v for k, v in ckpt['base_model'].items()}
        for k in list(base.keys()):
            for prefix in ('MAE_encoder.', 'base_model.'):
                if k.startswith(prefix[:-1]):
                    base[k[len(prefix):]] = base.pop(k)
                    break
        return self.load_state_dict(base, strict=False)

    def _level2(self, center):
        """Level-2 centres and index tensors for the prompt-propagation step (reference :631-643)."""
        if not self.config.prompt_propagation_after:
            return {}
        level2 = Group(num_group=self.num_group // 2, group_size=8)
        _, center2, center1_idx, center2_idx = level2(center, require_index=True, gather_idx=self.config.gather_idx)
        return dict(center1=center, center1_idx=center1_idx, center2=center2, center2_idx=center2_idx,
                    gather_idx=self.config.gather_idx, prompt_propagation_after=self.config.prompt_propagation_after)

    def _rectify(self, pts, point_num):
        """Denoising prompter (reference :541-570): score every input point, nudge the cloud by
        0.2 * predicted offset, keep the int(0.95 * point_num) least suspicious points."""
        grouper = Group(num_group=self.vis_num, group_size=16)
        neighborhood, vis_center = grouper(pts)
        tokens = self.encoder(neighborhood)
        pos = L.mlp2(self.pos_embed, vis_center)
        tokens = self.blocks(tokens, pos, path='rectify', rectify_adapter=True, rectify_prompts=True,
                             rectify_depth=self.config.prompter_config['rectify_depth'])
        return self.rectify_prompter.select(pts, vis_center, tokens, int(point_num * 0.95), nudge=0.2)

    def _complete(self, pts, point_num):
        """Completion prompter (reference :572-610): predict 32 missing centres and 32x32 points
        around them, append a quarter of them and re-sample to point_num."""
        grouper = Group(num_group=self.vis_num, group_size=16)
        neighborhood, vis_center = grouper(pts)
        _, rebuild = self._reconstruct(self.encoder(neighborhood), vis_center)
        self.aux['rebuild_points'] = rebuild          # (B, n_masked * group_size, 3): what a reconstruction loss would look at
        sampled, _ = misc.fps(rebuild, point_num // 4)
        if L.DIAG_AUX:
            self.aux['dbg_vis_center'], self.aux['dbg_sampled'] = vis_center, sampled
        pts = torch.cat([pts, sampled], dim=1).contiguous()
        if pts.shape[1] > point_num:
            pts = misc.fps(pts, point_num)[0]
        return pts

    def _reconstruct(self, tokens, vis_center):
        """Visible tokens -> (predicted centres of the missing groups (B,n_masked,3), rebuilt points (B,n_masked*group_size,3))
        through the pretask path of the backbone and the MAE decoder (reference :584-606, pretask_dev.py:713-737)."""
        B = tokens.shape[0]
        x_vis = tokens.reshape(B, -1, self.trans_dim)
        pos = L.mlp2(self.pos_embed, vis_center)
        x_vis = self.blocks(x_vis, pos, path='pretask', pretask_adapter=True, pretask_prompts=True,
                            pretask_depth=self.config.prompter_config['pretask_depth'])
        x_vis = HF.layer_norm(x_vis, self.norm)                    # (row kernel: no torch launch in the front-end)
        pos_vis = L.mlp2(self.decoder_pos_embed, vis_center).reshape(B, -1, self.trans_dim)
        shape_feature = L.mlp2(self.shape_pred, x_vis).reshape(B, self.vis_short * self.vis_num)
        predict_center = L.mlp2(self.coarse_pred, shape_feature).reshape(B, self.n_masked, 3)
        predict_token = L.mlp2(self.predict_token_generator, x_vis)
        pos_mask = L.mlp2(self.decoder_pos_embed, predict_center).reshape(B, -1, self.trans_dim)
        N = pos_mask.shape[1]
        mask_token = propagate(predict_center, vis_center, HF.expand_rows(self.mask_token, B, N), predict_token,
                               de_neighbors=6)
        x_rec = self.MAE_decoder(torch.cat([x_vis, mask_token], dim=1), torch.cat([pos_vis, pos_mask], dim=1), N,
                                 pretask_adapter=True, path='pretask')
        M = x_rec.shape[1]
        head = self.dense_pred[0]                                  # Conv1d(D, 3*group_size, 1) == a per-token Linear
        rel = HF.linear(x_rec, head.weight.squeeze(-1), head.bias).reshape(B, M, -1, 3)
        return predict_center, (rel + predict_center.unsqueeze(-2)).reshape(B, -1, 3)


@MODELS.register_module()
class Point_MAE_unify(PromptedBackbone):
    def __init__(self, config):
        super().__init__()
        D = self._build_backbone(config)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.cls_pos = nn.Parameter(torch.randn(1, 1, D))
        self.cls_head_finetune = nn.Sequential(
            nn.Linear(D * 2, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True), nn.Dropout(0.5),
            nn.Linear(256, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True), nn.Dropout(0.5),
            nn.Linear(256, self.cls_dim))
        for layer in self.cls_head_finetune:
            if isinstance(layer, nn.Linear):
                nn.init.kaiming_uniform_(layer.weight, a=math.sqrt(5.0))
        trunc_normal_(self.cls_token, std=.02)
        trunc_normal_(self.cls_pos, std=.02)
        self.build_loss_func()

    # ------------------------------------------------------------------ loss / checkpoints
    def build_loss_func(self):
        self.loss_ce = nn.CrossEntropyLoss()

    def get_loss_acc(self, ret, gt):
        if ret.is_cuda and ret.dtype == torch.float32 and ret.dim() == 2 and ret.shape[1] <= 512 and type(self.loss_ce) is nn.CrossEntropyLoss \
                and self.loss_ce.weight is None and self.loss_ce.label_smoothing == 0.0 and self.loss_ce.reduction == 'mean':
            return HF.cross_entropy_acc(ret, gt.long())        # loss, accuracy and d loss / d logits: one launch
        loss = self.loss_ce(ret, gt.long())
        acc = (ret.argmax(-1) == gt).sum() / float(gt.size(0))
        return loss, acc * 100

    def forward(self, pts, label=None, completion_prompt=False, denoise=False, point_num=1024, **kwargs):
        L.begin_forward(pts.device, self.training)
        try:
            return self._forward(pts, label=label, completion_prompt=completion_prompt, denoise=denoise, point_num=point_num, **kwargs)
        finally:
            L.end_forward()

    def prompt_points(self, pts, completion_prompt=True, denoise=True, point_num=1024):
        """The prompting front-end alone: raw (noisy / incomplete) clouds -> rectified + completed clouds (B, point_num, 3).
        `forward(prompt_points(pts), completion_prompt=False, denoise=False)` == `forward(pts, True, True)`."""
        L.begin_forward(pts.device, self.training)
        try:
            return self._prompt(pts, completion_prompt, denoise, point_num)
        finally:
            L.end_forward()

    def prompt_tokens(self, pts, completion_prompt=True, denoise=True, point_num=1024):
        """Everything of the forward that reads no PEFT-trainable parameter: the prompting front-end, grouping and patch
        embedding of the prompted cloud, positional embedding of the centres, the level-2 grouping and the index lists of
        the propagation step -> a flat tuple of tensors `state` (tokens (B,G,C), centres (B,G,3), ...).
        `forward_tokens(*prompt_tokens(pts))` == `forward(pts, True, True)`; a training step may run this for the NEXT batch
        while the trainable back-end works on the current one (upp_hip.train.PipelinedTrainStep)."""
        L.begin_forward(pts.device, self.training)
        try:
            return self._front_state(self._prompt(pts, completion_prompt, denoise, point_num))
        finally:
            L.end_forward()

    def forward_tokens(self, tokens, center, *rest):
        """The trainable back-end: the state produced by prompt_tokens -> logits."""
        L.begin_forward(tokens.device, self.training)
        try:
            return self._head(tokens, center, rest)
        finally:
            L.end_forward()

    def _front_state(self, pts):
        tokens, center = self._embed(pts)
        state = (tokens, center, L.mlp2(self.pos_embed, center))
        lvl2 = self._level2(center)
        if lvl2:
            state += (lvl2['center2'], lvl2['center1_idx'], lvl2['center2_idx'])
            prompts = getattr(self.blocks.blocks[0], 'downstream_prompts', None)
            if tokens.is_cuda and prompts is not None and lvl2['center2'].shape[1] <= 64:
                B, Lp = tokens.shape[0], 1 + tokens.shape[1] + prompts.shape[0]      # [cls | prompts | tokens] rows per sample
                entry = L.build_prop_index(center, lvl2['center2'], lvl2['center1_idx'], lvl2['center2_idx'],
                                           bool(self.config.gather_idx), B, Lp, 1)
                state += entry.tensors()
        return state

    def _prompt(self, pts, completion_prompt, denoise, point_num):
        if denoise:
            pts = self._rectify(pts, point_num)
        if completion_prompt:
            pts = self._complete(pts, point_num)
        return pts

    def _embed(self, pts):
        neighborhood, center = self.group_divider(pts)
        return self.encoder(neighborhood), center

    def _head(self, tokens, center, rest=()):
        B = tokens.size(0)
        x = torch.cat((self.cls_token.expand(B, -1, -1), tokens), dim=1)
        pos_tokens = rest[0] if len(rest) > 0 else L.mlp2(self.pos_embed, center)
        pos = torch.cat((self.cls_pos.expand(B, -1, -1), pos_tokens), dim=1)

        if len(rest) >= 4:         # level-2 grouping handed over by prompt_tokens
            propagation = dict(center1=center, center1_idx=rest[2], center2=rest[1], center2_idx=rest[3],
                               gather_idx=self.config.gather_idx, prompt_propagation_after=self.config.prompt_propagation_after)
            if len(rest) >= 14:    # ... and the propagation index lists for the [cls | prompts | tokens] layout
                prompts = self.blocks.blocks[0].downstream_prompts
                Lp = 1 + tokens.shape[1] + prompts.shape[0]
                propagation['_prop_entry'] = ((B, Lp, 1), HF.PropIndex.from_tensors(rest[4:14], rows=B * Lp))
        else:
            propagation = self._level2(center)
        if torch.is_grad_enabled() and self.cls_pos.requires_grad and not pos_tokens.requires_grad:
            # only row 0 of `pos` is trainable (PEFT): let the fused blocks route its gradient (see TransformerEncoder)
            propagation['cls_pos_param'] = self.cls_pos
        x = self.blocks(x, pos, path='downstream', downstream_adapter=True, downstream_prompts=True,
                        classification=True, **propagation)
        if (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] <= 512 and x.shape[1] >= 2 and isinstance(self.norm, nn.LayerNorm)
                and L.POOL_TRACE is None
                and not (torch.is_grad_enabled() and (self.norm.weight.requires_grad or self.norm.bias.requires_grad))):
            feat = HF.cls_pool(x, self.norm)                   # final LayerNorm + [cls | max over tokens]: one launch each way
        else:
            x = self.norm(x)
            feat = torch.cat([x[:, 0], L.max_over(x[:, 1:], 1, 'cls.max')], dim=-1)
        return self._cls_head(feat)

    def _cls_head(self, feat):
        """cls_head_finetune; on the HIP path every [BatchNorm1d, ReLU, Dropout] run after a Linear is one launch each way."""
        layers = list(self.cls_head_finetune)
        if L.sync_bn_active(self.training):               # (--sync_bn: batch statistics over all ranks, the tail unfused)
            x = feat
            for layer in self.cls_head_finetune:
                if isinstance(layer, nn.BatchNorm1d):
                    if self.training and layer.track_running_stats:
                        L.bump_counter(layer.num_batches_tracked)
                    x = L._bn_rows(x, layer, self.training)
                else:
                    x = HF.linear(x, layer.weight, layer.bias, own_wgrad=True) if isinstance(layer, nn.Linear) and x.is_cuda else layer(x)
            return x
        if not (feat.is_cuda and feat.dtype == torch.float32 and feat.dim() == 2):
            return self.cls_head_finetune(feat)
        x, i = feat, 0
        while i < len(layers):
            if (i + 2 < len(layers) and isinstance(layers[i], nn.BatchNorm1d) and isinstance(layers[i + 1], nn.ReLU)
                    and isinstance(layers[i + 2], nn.Dropout) and layers[i].affine):
                bn, drop = layers[i], layers[i + 2]
                p = drop.p if self.training else 0.0
                u = L.UNIFORMS.take(tuple(x.shape), x.device) if p > 0 else None
                if self.training and bn.track_running_stats:
                    L.bump_counter(bn.num_batches_tracked)
                x = HF.bn_relu_drop(x, bn, u, p, self.training)
                i += 3
            else:
                # (the head is trainable: data and weight gradients on upp_linear_f32 / upp_linear_wgrad_f32, 32 rows)
                x = HF.linear(x, layers[i].weight, layers[i].bias, own_wgrad=True) if isinstance(layers[i], nn.Linear) else layers[i](x)
                i += 1
        return x

    def _forward(self, pts, label=None, completion_prompt=False, denoise=False, point_num=1024, **kwargs):
        return self._head(*self._embed(self._prompt(pts, completion_prompt, denoise, point_num)))
