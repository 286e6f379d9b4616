"""Building blocks of the UPP point-cloud transformer, restated for the MI355X path.

Parameter names and shapes reproduce the reference state-dict schema so
that reference checkpoints load; the forward code is organised around the fused gfx950
operators (FPS+gather, kNN+group) and channels-last GEMM views instead of the reference's
Conv1d/transposes.  file:line citations point into the reference repository.
"""
import math

import os
import torch
import torch.nn as nn
import torch.nn.functional as F

from upp_hip import functional as HF

# Operator table for the two grouping primitives.  Product code always runs the HIP
# operators; tests on a GPU-less host swap in oracle-backed callables here (test
# injection only -- nothing in this package imports the oracle).
OPS = {
    "fps_gather": HF.fps_gather,   # (xyz (B,N,3), G)          -> centers (B,G,3), idx (B,G) int32
    "knn_group": HF.knn_group,     # (xyz, centers (B,G,3), k) -> neigh (B,G,k,3), idx (B,G,k) int64
}


# --------------------------------------------------------------------------- small helpers
class UniformBank:
    """One U[0,1) draw per model forward, sliced by the fused blocks' drop-path / dropout sites.

    A training step of the classification model asks for ~50 small uniform tensors; as separate
    torch.rand launches they cost ~5 us each on MI355X.  `begin()` (called at the top of a model
    forward) draws as many values as the previous forward consumed; `take()` hands out views and
    falls back to a direct draw while the demand is still unknown (first forward) or grew."""

    def __init__(self, generator=None):
        self.buf, self.pos, self.asked, self.need = None, 0, 0, 0
        self.generator = generator          # None = the device's default generator

    def begin(self, device, training):
        self.need = max(self.need, self.asked)
        self.pos = self.asked = 0
        self.buf = torch.rand(self.need, device=device, generator=self.generator) if (training and self.need) else None

    def take(self, shape, device):
        n = math.prod(shape)
        self.asked += n
        if self.buf is not None and self.buf.device == device and self.pos + n <= self.buf.numel():
            out = self.buf[self.pos:self.pos + n].view(shape)
            self.pos += n
            return out
        return torch.rand(shape, device=device, generator=self.generator)


UNIFORMS = UniformBank()
FUSE_INTERP_GEO = True     # _inverse_distance_interp: HF.interp_geo when the coordinates carry a gradient (False: the torch formulation; tests A/B it)
FUSE_LN_ADAPTER = True     # Block.forward_fused: close the block with HF.ln_adapter (one launch) instead of HF.rowln + HF.adapter


class use_rng:
    """Scope in which the model's random draws (the uniform bank, DropPath) come from `bank` -- a UniformBank with its own
    torch.Generator.  PipelinedTrainStep runs the front-end under one: its graphs replay on a second stream beside the
    back-end's, and two concurrently replaying graphs that captured draws from the SAME generator race on its philox offset
    (and a shared bank would make the front-end draw the back-end's demand)."""

    def __init__(self, bank):
        self.bank = bank

    def __enter__(self):
        global UNIFORMS
        self.prev, UNIFORMS = UNIFORMS, self.bank
        return self.bank

    def __exit__(self, *exc):
        global UNIFORMS
        UNIFORMS = self.prev
        return False

# BatchNorm `num_batches_tracked` counters touched during one model forward: bumped together by ONE multi-tensor
# launch at the end of the forward instead of one tiny kernel per BatchNorm call.
_pending_counters = None


def bump_counter(t):
    """`num_batches_tracked += 1` of a BatchNorm the caller is about to run in training mode -- THE one place a counter moves: every call
    site of _bn_rows / the fused BatchNorm kernels bumps through here, the layer functions themselves never do.  Under synchronised
    BatchNorm the bump is immediate (momentum=None reads the counter inside the layer); otherwise it joins the forward's batched bump."""
    if _pending_counters is None or sync_bn_active():
        t.add_(1)
    else:
        _pending_counters.append(t)


def begin_forward(device, training):
    """Called at the top of a model forward: draws the forward's uniforms, opens the counter list."""
    global _pending_counters
    UNIFORMS.begin(device, training)
    _pending_counters = []


def end_forward():
    global _pending_counters
    pending, _pending_counters = _pending_counters, None
    if pending:
        # a module called several times per forward (the shared patch embedding) appears several times: one entry per
        # tensor with its count -- duplicates inside one multi-tensor launch would race and lose increments
        uniq, count = {}, {}
        for t in pending:
            uniq[t.data_ptr()] = t
            count[t.data_ptr()] = count.get(t.data_ptr(), 0) + 1
        keys = list(uniq)
        torch._foreach_add_([uniq[k] for k in keys], [count[k] for k in keys])


class DropPath(nn.Module):
    """timm 0.4.5 DropPath: per-sample stochastic depth, identity in eval mode."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = drop_prob

    def sample_scale(self, x):
        """Per-sample factor floor(keep + U) / keep, or None when the layer is the identity."""
        if self.drop_prob == 0. or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        return x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).uniform_(generator=UNIFORMS.generator).add_(keep).floor_().div_(keep)

    def forward(self, x):
        scale = self.sample_scale(x)
        return x if scale is None else x * scale


def trunc_normal_(t, std=.02):
    return nn.init.trunc_normal_(t, std=std)


def square_distance(src, dst):
    """|s|^2 + |d|^2 - 2 s.d^T, (B,N,C) x (B,M,C) -> (B,N,M)  (models/modules.py:13-32)."""
    if src.is_cuda and src.shape[-1] == 3 and src.shape[0] * src.shape[1] * dst.shape[1] <= (1 << 20):
        # a few thousand pairs of 3-vectors (the prompters' centre sets when the geometry carries a gradient): the products as element-wise
        # ops -- a batched library GEMM with K = 3 plus its two backward GEMMs otherwise; same formula, the dot product summed x, y, z
        dist = -2 * (src.unsqueeze(2) * dst.unsqueeze(1)).sum(-1)
    else:
        dist = -2 * torch.matmul(src, dst.transpose(1, 2))
    dist += torch.sum(src ** 2, -1).unsqueeze(-1)
    dist += torch.sum(dst ** 2, -1).unsqueeze(1)
    return dist


def index_points(points, idx):
    """points (B,N,C), idx (B,...) -> (B,...,C)  (models/modules.py:35-51)."""
    B = points.shape[0]
    batch = torch.arange(B, dtype=torch.long, device=points.device).view((B,) + (1,) * (idx.dim() - 1))
    return points[batch.expand_as(idx), idx, :]


def _inverse_distance_interp(xyz1, xyz2, points2, k, eps, out=None, col0=0):
    """k nearest of xyz2 for each xyz1 point (full sort, as the reference), 1/(d+eps) weights.
    `out`/`col0`: optional (B,N,W) buffer whose columns [col0, col0+C) receive the result (fused path only)."""
    S = xyz2.shape[1]
    if POOL_TRACE is not None and out is None:      # (test instrument: the torch formulation on the traced neighbour lists)
        sq = square_distance(xyz1, xyz2)
        idx = trace_idx('interp.knn', HF.argsort_rows(sq)[:, :, :k])
        recip = 1.0 / (sq.gather(-1, idx) + eps)
        return torch.sum(index_points(points2, idx) * (recip / torch.sum(recip, dim=2, keepdim=True)).unsqueeze(-1), dim=2)
    if (out is None and xyz1.is_cuda and xyz1.dtype == torch.float32 and xyz2.dtype == torch.float32 and points2.dtype == torch.float32
            and S <= 256 and min(k, S) <= 16 and xyz1.shape[-1] == 3 and not _no_grad_needed(xyz1, xyz2) and FUSE_INTERP_GEO):
        # the geometry carries a gradient (stage 2, the pre-task recipe): one autograd node for table + interpolation + all three gradients
        return HF.interp_geo(xyz1, xyz2, points2, k, eps)
    if (xyz1.is_cuda and xyz1.dtype == torch.float32 and xyz2.dtype == torch.float32 and S <= 256 and min(k, S) <= 16
            and xyz1.shape[-1] == 3 and _no_grad_needed(xyz1, xyz2)):
        dists, idx = HF.sqdist_topk(xyz1, xyz2, min(k, S))       # one launch for matmul + 5 element-wise passes + full sort
    else:
        sq = square_distance(xyz1, xyz2)                      # (the reference's full sort: rank-counting kernel, no library sort)
        idx = HF.argsort_rows(sq)
        dists = sq.gather(-1, idx)
    if points2.is_cuda and points2.dtype == torch.float32 and k <= 16:
        if _no_grad_needed(xyz1, xyz2, points2):
            return HF.interp(dists, idx, points2, min(k, dists.shape[-1]), eps, out, col0)
        if out is None and _no_grad_needed(xyz1, xyz2) and xyz1.shape[1] <= 4096:     # trainable features, constant geometry
            return HF.interp_train(dists, idx, points2, min(k, dists.shape[-1]), eps)
    assert out is None
    dists, idx = dists[:, :, :k], idx[:, :, :k]
    recip = 1.0 / (dists + eps)
    weight = recip / torch.sum(recip, dim=2, keepdim=True)
    return torch.sum(index_points(points2, idx) * weight.unsqueeze(-1), dim=2)


def _prop_lists_torch(c1, c2, i1, i2, gather_idx, B, Lp, off):
    """Index lists of the fused propagation step with the reference's torch ops (what upp_prop_index computes in one
    launch): absolute rows for the neighbour / centre indices and the 8 nearest level-2 centres with their weights."""
    sq = square_distance(c1, c2)
    idx8 = trace_idx('prop.knn8', HF.argsort_rows(sq)[:, :, :8])
    d8 = sq.gather(-1, idx8)
    recip = 1.0 / (d8 + 1e-3)
    w8 = (recip / torch.sum(recip, dim=2, keepdim=True)).contiguous()
    if gather_idx:
        base = (torch.arange(B, device=c1.device) * Lp + off).view(B, 1)
        i1a, i2a = base + i1.reshape(B, -1), base + i2.reshape(B, -1)
    else:
        G = Lp - off      # the reference indexes the cls-stripped tokens as a flat (B*G)-row matrix
        i1a = torch.div(i1, G, rounding_mode='floor') * Lp + off + i1 % G
        i2a = torch.div(i2, G, rounding_mode='floor') * Lp + off + i2 % G
    return i1a.reshape(-1).int().contiguous(), i2a.reshape(-1).int().contiguous(), idx8.int().contiguous(), w8


def build_prop_index(c1, c2, i1, i2, gather_idx, B, Lp, off):
    """HF.PropIndex of the fused propagation step for a (B, Lp)-row token matrix with `off` leading rows per sample."""
    G2 = c2.shape[1]
    with torch.no_grad():
        if G2 <= 64 and i1.dtype == torch.int64 and i2.dtype == torch.int64 and POOL_TRACE is None:
            lists = HF.ops.prop_index(c1.contiguous(), c2.contiguous(), i1.contiguous(), i2.contiguous(), gather_idx, Lp, off, 1e-3)
        else:
            lists = _prop_lists_torch(c1, c2, i1, i2, gather_idx, B, Lp, off)
        return HF.PropIndex(*lists, rows=B * Lp)


def propagate(xyz1, xyz2, points1, points2, de_neighbors=64, dist_e=1e-8):
    """points1 + 0.3 * inverse-distance interpolation of points2  (models/Point_MAE_unify.py:22-48)."""
    return points1 + 0.3 * _inverse_distance_interp(xyz1, xyz2, points2, de_neighbors, dist_e)


DIAG_AUX = bool(os.environ.get('UPP_DIAG_AUX'))       # (diagnostic: models keep front-end intermediates in .aux -- tools/micro/pipe_race_probe.py)
POOL_TRACE = None          # test instrument (tests/test_gpu_model.py): {'mode': 'record' | 'replay', 'items': [...]} -- see max_over


def max_over(x, dim, site=''):
    """x.max(dim)[0] at the max-pool sites that stay autograd ops.  With POOL_TRACE set, the arg-max of every call is recorded
    ('record') or TAKEN from the trace instead of being recomputed ('replay') -- a float64 evaluation can then be given the gates an f32
    evaluation chose, and the two compared without arg-max flips between nearly equal candidates (round-3 verdict, item 6).  A recorded
    item is consumed by the next call of the same site and input shape; calls the recording run did not make (branches it ran on fused
    kernels) compute their own arg-max."""
    t = POOL_TRACE
    if t is None:
        if x.is_cuda and dim % x.dim() == x.dim() - 2:
            return HF.group_max(x)          # (max + arg-max kernel, one-pass backward; torch: reduce, then zero-fill + scatter)
        return x.max(dim=dim)[0]
    key = (site, tuple(x.shape), dim)
    if t['mode'] == 'record':
        v, i = x.max(dim=dim)
        t['items'].append((key, i.detach().cpu()))
        return v
    pos = t.setdefault('pos', 0)
    if pos < len(t['items']) and t['items'][pos][0] == key:
        t['pos'] = pos + 1
        return x.gather(dim, t['items'][pos][1].to(x.device).unsqueeze(dim)).squeeze(dim)
    return x.max(dim=dim)[0]


def gate(site, z, slope=0.0):
    """relu(z) -- leaky_relu(z, slope) for slope > 0 -- at the ReLU sites that are taken as plain torch ops while POOL_TRACE is set (the same
    instrument as max_over / trace_idx: every fused kernel that applies a ReLU declines under POOL_TRACE).  'record': the mask z > 0 is
    kept; 'replay': the RECORDED mask gates z, so that an f32 and an f64 evaluation of the segmentation step -- whose train-mode BatchNorm
    puts thousands of pre-activations within 1e-7 of zero -- differ by rounding only, not by a redraw of the gates (round-4 verdict 6d)."""
    t = POOL_TRACE
    plain = lambda: F.leaky_relu(z, slope) if slope else F.relu(z)          # noqa: E731
    if t is None:
        return plain()
    key = (site, tuple(z.shape), -2)
    if t['mode'] == 'record':
        t['items'].append((key, (z > 0).detach().cpu()))
        return plain()
    pos = t.setdefault('pos', 0)
    if pos < len(t['items']) and t['items'][pos][0] == key:
        t['pos'] = pos + 1
        m = t['items'][pos][1].to(z.device)
        return z * torch.where(m, torch.ones((), dtype=z.dtype, device=z.device), torch.full((), slope, dtype=z.dtype, device=z.device))
    return plain()


def trace_idx(site, idx):
    """The same instrument for the other discrete choices of a forward (FPS picks, neighbour lists, the rectify prompter's ranking): the
    index tensor is recorded, or REPLACED by the recorded one.  Callers recompute whatever they derive from the indices."""
    t = POOL_TRACE
    if t is None:
        return idx
    key = (site, tuple(idx.shape), -1)
    if t['mode'] == 'record':
        t['items'].append((key, idx.detach().cpu()))
        return idx
    pos = t.setdefault('pos', 0)
    if pos < len(t['items']) and t['items'][pos][0] == key:
        t['pos'] = pos + 1
        return t['items'][pos][1].to(device=idx.device, dtype=idx.dtype)
    return idx


def pooling(x, transform):
    """(B,G,k,C) -> (B,G,C).  `pooling` is called at models/Point_MAE_pretask_dev.py:294 but defined
    nowhere in the reference; this is the Point-PEFT form the README credits:
    max + mean over the neighbourhood, then the block's BatchNorm1d over channels.  ASSUMPTION."""
    lc = max_over(x, 2, 'block.pooling') + x.mean(dim=2)
    if isinstance(transform, nn.BatchNorm1d) and sync_bn_active(transform.training):
        if transform.track_running_stats and transform.num_batches_tracked is not None:
            bump_counter(transform.num_batches_tracked)
        return _bn_rows(lc.reshape(-1, lc.shape[-1]), transform, transform.training).view(lc.shape)
    return transform(lc.permute(0, 2, 1)).permute(0, 2, 1)


# --------------------------------------------------------------------------- grouping
_OFFSETS = {}


def _batch_offsets(B, N, device):
    """(B,1,1) int64 tensor b*N, built once per (B, N, device): a constant, not two launches per grouping call."""
    key = (B, N, str(device))
    if key not in _OFFSETS:
        fresh = torch.arange(B, device=device).view(-1, 1, 1) * N
        if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            return fresh        # memory of a graph's private pool, filled only on replay: never cache it
        if len(_OFFSETS) > 64:
            _OFFSETS.clear()
        _OFFSETS[key] = fresh
    return _OFFSETS[key]


class Group(nn.Module):
    """FPS centres + kNN neighbourhoods, centred  (models/Point_MAE_unify.py:51-92).

    One FPS launch (indices + centre coordinates) and one kNN launch (indices + centred
    neighbourhood) replace the reference's FPS, gather, 3*B kNN launches, 2 aranges, flat
    gather and subtraction."""

    def __init__(self, num_group, group_size):
        super().__init__()
        self.num_group = num_group
        self.group_size = group_size

    def forward(self, xyz, require_index=False, gather_idx=False):
        B, N, _ = xyz.shape
        xyz = xyz.contiguous()
        center, center_idx = OPS["fps_gather"](xyz, self.num_group)
        neighborhood, idx = OPS["knn_group"](xyz, center, self.group_size)
        if POOL_TRACE is not None:
            center_idx, idx = trace_idx('group.fps', center_idx), trace_idx('group.knn', idx)
            center = index_points(xyz, center_idx.long())
            neighborhood = index_points(xyz, idx) - center.unsqueeze(2)
        if not require_index:
            return neighborhood, center
        if not gather_idx:
            # flat indices into a (B*N, C) view, as the reference hands them on (:73-79)
            base = _batch_offsets(B, N, xyz.device)
            idx = (idx + base).view(-1)
            center_idx = (center_idx + base.view(-1, 1)).view(-1)
        else:
            center_idx = center_idx.long()
        return neighborhood, center, idx, center_idx


# --------------------------------------------------------------------------- patch embedding
def _frozen_bias(linear):
    """A Linear whose bias exists and needs no gradient: its bias can be added by the kernel that consumes the GEMM output."""
    return linear.bias is not None and not (torch.is_grad_enabled() and linear.bias.requires_grad)


def _no_grad_needed(*tensors):
    """True when no autograd graph has to be recorded for an op on these tensors (frozen branch / no_grad)."""
    return not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors))


# ---- optional synchronised BatchNorm (reference tools/runner_module.py:50-52, tools/runner_unify_seg.py:129-131: `--sync_bn` converts every
# BatchNorm to torch.nn.SyncBatchNorm before the DDP wrap).  Off by default, as in the reference's recipes.  With the switch on and
# torch.distributed initialised, every training-mode BatchNorm of the model takes its statistics over the rows of ALL ranks: the per-rank
# (count, sum, sum of squares) of a layer travel in one all-reduce, the backward's two column sums in another; the fused kernels that
# compute batch statistics on the device (patch-embedding chain, prompt propagation, classification-head tail) decline while it is on.
SYNC_BN = False


def enable_sync_bn(on=True):
    global SYNC_BN
    SYNC_BN = bool(on)


def sync_bn_active(training=True):
    import torch.distributed as dist
    return SYNC_BN and training and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class _SyncBNRows(torch.autograd.Function):
    """y = (x - mean) * rstd * gamma + beta over the rows of every rank (biased variance for the output, unbiased for running_var: nn.BatchNorm)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, counter=None):
        # The row count stays ON THE DEVICE (a 1-element tensor): a .item() here is a host sync, which a HIP-graph capture refuses -- and the
        # multi-GPU step drivers this switch exists for capture their step.  Sums in float64 (one pass; cancellation-free for E[x^2] - mean^2),
        # the normalised rows in the input's precision.
        import torch.distributed as dist
        C = x.shape[1]
        xd = x.double()
        stats = torch.cat([xd.sum(0), (xd * xd).sum(0), torch.full((1,), float(x.shape[0]), dtype=torch.float64, device=x.device)])
        dist.all_reduce(stats)
        n = stats[2 * C:2 * C + 1]
        mean = stats[:C] / n
        var = (stats[C:2 * C] / n - mean * mean).clamp_min(0.0)
        rstd = (var + eps).rsqrt()
        if running_mean is not None:
            with torch.no_grad():
                unbiased = var * (n / (n - 1.0).clamp_min(1.0))
                if momentum is None:            # cumulative moving average, as nn.BatchNorm: factor 1 / num_batches_tracked (already bumped)
                    f = (1.0 / counter.double().clamp_min(1.0)).to(running_mean.dtype)
                    running_mean.mul_(1 - f).add_(mean.to(running_mean.dtype) * f)
                    running_var.mul_(1 - f).add_(unbiased.to(running_var.dtype) * f)
                else:
                    running_mean.mul_(1 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
                    running_var.mul_(1 - momentum).add_(unbiased.to(running_var.dtype), alpha=momentum)
        mean32, rstd32 = mean.to(x.dtype), rstd.to(x.dtype)
        xhat = (x - mean32) * rstd32
        ctx.save_for_backward(xhat, weight, rstd32, n)
        y = xhat if weight is None else xhat * weight
        return y if bias is None else y + bias

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        xhat, weight, rstd, n = ctx.saved_tensors
        C = g.shape[1]
        gd = g.double()
        sums = torch.cat([gd.sum(0), (gd * xhat.double()).sum(0)])
        local = sums.clone()
        dist.all_reduce(sums)
        mg, mgx = (sums[:C] / n).to(g.dtype), (sums[C:] / n).to(g.dtype)
        scale = rstd if weight is None else rstd * weight
        gx = (g - mg - xhat * mgx) * scale if ctx.needs_input_grad[0] else None
        # rank-local sums for the parameters: the step driver's gradient all-reduce (utils.dist_utils.FlatGradAllReduce, an average of the
        # ranks' buffers as DistributedDataParallel's) then yields what nn.SyncBatchNorm + DDP yield
        gw = local[C:].to(g.dtype) if weight is not None and ctx.needs_input_grad[1] else None
        gb = local[:C].to(g.dtype) if ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None, None, None, None


def _bn_rows(x, bn, training, relu=False, drop=None):
    """BatchNorm1d of a channels-last (rows, C) matrix: identical statistics to BatchNorm1d on
    the reference's (BG, C, n) layout (both reduce over every position of every group).
    drop: an nn.Dropout that follows (the segmentation head's `BatchNorm1d, ReLU, Dropout(0.5)`): applied in the BatchNorm's own passes on
    the GPU training path (masks from bn.num_batches_tracked, which the caller has bumped for this forward), by the module elsewhere."""
    if drop is not None:
        p = float(drop.p) if (drop.training and training) else 0.0
        fused = (p > 0.0 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and POOL_TRACE is None and bn.track_running_stats
                 and bn.num_batches_tracked is not None and bn.momentum is not None and not _no_grad_needed(x, bn.weight, bn.bias)
                 and not sync_bn_active(True) and not drop.inplace)
        if fused:
            return HF.bn_rows_train(x, bn, relu, drop_p=p, salt=x.shape[1], bump_pending=_pending_counters is not None)
        return drop(_bn_rows(x, bn, training, relu))
    if POOL_TRACE is not None and relu and x.dim() == 2 and not sync_bn_active(training or bn.running_mean is None):
        # test instrument: the BatchNorm on our kernels (or torch, on the host), the ReLU as a recorded / replayed gate
        return gate('bn_rows.relu', _bn_rows(x, bn, training, relu=False))
    if x.dim() == 2 and sync_bn_active(training or bn.running_mean is None):
        # (the caller has bumped num_batches_tracked already -- bump_counter is immediate while this switch is on: momentum=None reads it)
        y = _SyncBNRows.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, bn.num_batches_tracked)
        return F.relu(y) if relu else y
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
        if _no_grad_needed(x, bn.weight, bn.bias):
            return HF.bn_rows(x, bn, training, relu)   # 3 launches instead of torch's 5-6 (frozen prompter branches)
        if (training or bn.running_mean is None) and bn.momentum is not None:
            return HF.bn_rows_train(x, bn, relu)       # trainable heads: batch statistics, own backward (3 + 3 launches)
    y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                     training, 0.0 if bn.momentum is None else bn.momentum, bn.eps)
    return F.relu(y) if relu else y


def mlp2(seq, x):
    """`Sequential(Linear, GELU, Linear)` -- the position MLPs (3 -> 128 -> D) and the prompter heads -- on HF.linear: the GELU rides
    in the first layer's epilogue, K = 3 takes the small-K kernel (reference models/Point_MAE_pretask_dev.py:395-399, 46)."""
    if not (x.is_cuda and x.dtype == torch.float32 and len(seq) == 3 and isinstance(seq[1], nn.GELU) and seq[1].approximate == 'none'):
        return seq(x)
    l0, l2 = seq[0], seq[2]
    if torch.is_grad_enabled() and (x.requires_grad or any(p is not None and p.requires_grad for p in (l0.weight, l0.bias, l2.weight, l2.bias))):
        # with gradients (stage 2, the pre-task and pre-training recipes): GELU' is saved by the first launch and multiplied in the second
        # layer's data-gradient epilogue -- no torch gelu / gelu_backward pair
        if HF.mlp_smallk_gelu_usable(x, l0.weight, l0.bias, l2.weight, l2.bias):
            return HF.mlp_smallk_gelu(x, l0.weight, l0.bias, l2.weight, l2.bias)
        if (HF.linear_usable(x, l0.weight) and l0.bias is not None and l0.out_features % 4 == 0 and l2.weight.shape[0] % 4 == 0
                and HF.linear_usable(x.new_empty((1, l0.out_features)), l2.weight)):
            return HF.mlp_gelu(x, l0.weight, l0.bias, l2.weight, l2.bias)
    return HF.linear(HF.linear(x, l0.weight, l0.bias, act='gelu'), l2.weight, l2.bias)


def _relu_linear(x, lin):
    """relu(lin(x)): the ReLU in the GEMM's epilogue; as a recorded / replayed gate while the test instrument is on."""
    if POOL_TRACE is not None:
        return gate('score_head.relu', HF.linear(x, lin.weight, lin.bias))
    return HF.linear(x, lin.weight, lin.bias, act='relu')


def _pointwise_bn_relu(x, conv, bn, training):
    """relu(bn(conv1x1(x))) on a channels-last (rows, C_in) matrix: a 1x1 Conv1d/Conv2d is a GEMM and
    BatchNorm over (batch, positions) is BatchNorm over rows -- no NCHW permutes, no MIOpen conv."""
    if training and bn.track_running_stats:
        bump_counter(bn.num_batches_tracked)
    w = conv.weight.view(conv.weight.shape[0], -1)
    return _bn_rows(HF.linear(x, w, conv.bias), bn, training, relu=True)


class Encoder(nn.Module):
    """mini-PointNet patch embedding  (models/Point_MAE_unify.py:191-222).

    Same parameters as the reference (Conv1d 1x1 weights (out,in,1)); evaluated as row-major
    GEMMs over all B*G*n points.  The 512->512 layer acts on cat([global, local]): the global
    half is constant over the n points of a group, so it is multiplied once per group
    (BG rows) and broadcast instead of n times."""

    def __init__(self, encoder_channel):
        super().__init__()
        self.encoder_channel = encoder_channel
        self.first_conv = nn.Sequential(
            nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True), nn.Conv1d(128, 256, 1))
        self.second_conv = nn.Sequential(
            nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Conv1d(512, encoder_channel, 1))

    def _fusable(self, x):
        """The gfx950 kernel chain is forward-only: usable whenever no gradient has to flow through the
        encoder (it is frozen in every UPP recipe), for the reference's layer sizes and group sizes."""
        if not x.is_cuda or x.dtype != torch.float32 or x.shape[2] not in (16, 32) or sync_bn_active(self.training):
            return False
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False
        c1, bn1, _, c2 = self.first_conv
        c3, bn3, _, c4 = self.second_conv
        return (tuple(c1.weight.shape) == (128, 3, 1) and tuple(c2.weight.shape) == (256, 128, 1)
                and tuple(c3.weight.shape) == (512, 512, 1) and c4.weight.shape[1] == 512 and c4.weight.shape[0] % 4 == 0
                and bn1.track_running_stats and bn3.track_running_stats and bn1.momentum == bn3.momentum and bn1.eps == bn3.eps)

    def forward(self, point_groups):
        if self._fusable(point_groups):
            from upp_hip import ops
            bn1, bn3 = self.first_conv[1], self.second_conv[1]
            if self.training:
                bump_counter(bn1.num_batches_tracked)
                bump_counter(bn3.num_batches_tracked)
            return ops.patch_embed_fwd(point_groups.contiguous(), self, self.training)
        return self._forward_torch(point_groups)

    def _w1_k32(self):
        """Conv1d(3,128) weight zero-padded to K = 32 (the contraction granule of upp_linear_f32); cached while frozen."""
        w = self.first_conv[0].weight
        if torch.is_grad_enabled() and w.requires_grad:
            return F.pad(w.squeeze(-1), (0, 29))
        key = (w.data_ptr(), w._version)
        buf = getattr(self, '_w1p', None)
        if buf is None or buf.device != w.device or buf.dtype != w.dtype:
            self._w1p, self._w1p_key = F.pad(w.detach().squeeze(-1), (0, 29)).contiguous(), key
            self._w1p._upp_persistent = True       # (ops.PLANES may cache the bf16 plane image of this buffer)
        elif self._w1p_key != key and not (w.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.refresh_padded_weight()
        return self._w1p

    def refresh_padded_weight(self):
        """Re-copy the frozen first-conv weight into its zero-padded K = 32 image IN PLACE: a captured step (and the W^T copy a
        data-gradient GEMM keeps of it, functional.TRANSPOSED) holds the buffer's address, so after load_state_dict into an already
        captured model the contents must change, never the storage.  functional.refresh_caches(model) calls this for every encoder."""
        w = self.first_conv[0].weight
        if getattr(self, '_w1p', None) is not None and self._w1p.device == w.device:
            with torch.no_grad():
                self._w1p[:, :w.shape[1]].copy_(w.detach().squeeze(-1))
            self._w1p_key = (w.data_ptr(), w._version)

    def _forward_torch(self, point_groups):
        """The differentiable formulation (a gradient reaches the encoder: Point-MAE pre-training trains it, stage 2 of the UPP
        recipe differentiates THROUGH it).  On the GPU every GEMM, forward and backward, is one of this library's kernels:
        upp_linear_f32 (outputs and data gradients; the K = 3 first layer zero-padded to the 32-wide contraction granule),
        upp_linear_wgrad_f32 (weight gradients), the BatchNorm(+ReLU) row kernels with their backward; the two max-pools
        and the group broadcast stay autograd ops.  Reference models/Point_MAE_unify.py:204-222."""
        bs, g, n, _ = point_groups.shape
        c1, bn1, _, c2 = self.first_conv
        c3, bn3, _, c4 = self.second_conv
        x = point_groups.reshape(bs * g * n, 3)
        if x.is_cuda and x.dtype == torch.float32:
            h = HF.linear(F.pad(x, (0, 29)), self._w1_k32(), c1.bias, own_wgrad=True)
        else:
            h = F.linear(x, c1.weight.squeeze(-1), c1.bias)
        if self.training and bn1.track_running_stats:
            bump_counter(bn1.num_batches_tracked)
            bump_counter(bn3.num_batches_tracked)
        h = _bn_rows(h, bn1, self.training, relu=True)
        f = HF.linear(h, c2.weight.squeeze(-1), c2.bias, own_wgrad=True)                    # (BGn, 256)
        fg = max_over(f.view(bs * g, n, 256), 1, 'encoder.pool1')           # (BG, 256)
        w3 = c3.weight.squeeze(-1)                                          # (512, 512): [global | local]
        hg = HF.linear(fg, w3[:, :256], c3.bias, own_wgrad=True)                            # (BG, 512) once per group (column windows: no copies)
        if (n & (n - 1)) == 0 and n >= 32:
            h = HF.linear_group_bias(f, w3[:, 256:], hg, n)                  # (the group's half: a bias per n rows in the GEMM's epilogue)
        else:
            h = HF.linear(f, w3[:, 256:], own_wgrad=True).view(bs * g, n, 512) + hg.unsqueeze(1)
        h = _bn_rows(h.view(bs * g * n, 512), bn3, self.training, relu=True)
        out = HF.linear(h, c4.weight.squeeze(-1), c4.bias, own_wgrad=True)                  # (BGn, C)
        return max_over(out.view(bs * g, n, self.encoder_channel), 1, 'encoder.pool2').view(bs, g, self.encoder_channel)


# --------------------------------------------------------------------------- transformer
class Mlp(nn.Module):
    """models/Point_MAE_pretask_dev.py:153-169"""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        fc1, fc2 = self.fc1, self.fc2
        if (self.drop.p == 0 and isinstance(self.act, nn.GELU) and self.act.approximate == 'none' and fc1.bias is not None
                and HF.linear_usable(x, fc1.weight) and fc1.out_features % 4 == 0):
            # bias + GELU (+ GELU' for backward) in the fc1 epilogue, GELU' applied in the epilogue of fc2's data gradient
            return HF.mlp_gelu(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias)
        h = self.drop(self.act(HF.linear(x, self.fc1.weight, self.fc1.bias)))
        return self.drop(HF.linear(h, self.fc2.weight, self.fc2.bias))


class Attention(nn.Module):
    """models/Point_MAE_pretask_dev.py:172-196: qkv (no bias), softmax(q k^T * d^-0.5) v, proj."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def fusable(self, x):
        """The gfx950 attention kernel serves head_dim 64, L <= 144 and no attention dropout (all UPP configs)."""
        return (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] // self.num_heads == 64 and x.shape[1] <= 144
                and not (self.training and self.attn_drop.p > 0))

    def forward(self, x):
        B, N, C = x.shape
        if self.fusable(x):
            ctx = HF.attention(HF.linear(x, self.qkv.weight, self.qkv.bias), self.num_heads, self.scale)
            return self.proj_drop(HF.linear(ctx, self.proj.weight, self.proj.bias))
        qkv = self.qkv(x).view(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = self.attn_drop(((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1))
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class Adapter(nn.Module):
    """0.7 * ln2(drop(GELU(ln1(LayerNorm(x)))))  (models/Point_MAE_pretask_dev.py:54-104; the
    variant without the `scale` Linear is the one Block instantiates)."""

    def __init__(self, embed_dims, reduction_dims, drop_rate_adapter=0.1):
        super().__init__()
        self.embed_dims = embed_dims
        self.super_reductuion_dim = reduction_dims
        self.dropout = nn.Dropout(p=drop_rate_adapter)
        if reduction_dims > 0:
            self.layer_norm = nn.LayerNorm(embed_dims)
            self.ln1 = nn.Linear(embed_dims, reduction_dims)
            self.activate = nn.GELU()
            self.ln2 = nn.Linear(reduction_dims, embed_dims)
            for m in (self.ln1, self.ln2):
                nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
                nn.init.normal_(m.bias, std=1e-6)

    def forward(self, x):
        return self.ln2(self.dropout(self.activate(self.ln1(self.layer_norm(x))))) * 0.7


_PATHS = ("rectify", "pretask", "downstream")


class Block(nn.Module):
    """Transformer block with per-path prompts / adapters and the prompt-propagation step
    (models/Point_MAE_pretask_dev.py:199-321)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, **kwargs):
        super().__init__()
        self.dim = dim
        self.norm1 = norm_layer(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                              attn_drop=attn_drop, proj_drop=drop)
        self.bnorm = nn.BatchNorm1d(dim)
        bi = kwargs.get('block_idx', 0)
        for path in _PATHS:   # construction order rectify -> pretask -> downstream, as the reference (:213-242)
            adapter = prompts = None
            want_adapter = kwargs.get(f'{path}_adapter', False)
            if want_adapter and (path == 'downstream' or bi < kwargs[f'{path}_depth']):
                adapter = Adapter(embed_dims=dim, reduction_dims=32, drop_rate_adapter=0.1)
            if kwargs.get(f'{path}_prompts', False) and bi < kwargs[f'{path}_prompts_depth']:
                prompts = nn.Parameter(torch.zeros(kwargs[f'{path}_prompts_num'], dim))
                nn.init.xavier_uniform_(prompts)
            setattr(self, f'{path}_adapter', adapter)
            setattr(self, f'{path}_adapter1', None)
            setattr(self, f'{path}_prompts', prompts)

    # -- prompt propagation (:275-303), with the reference's index semantics kept verbatim ------
    def _propagate_prompts(self, x, kw):
        is_cls = bool(kw.get('classification'))
        B, G, _ = x.shape
        if is_cls:
            cls_x, x = x[:, 0:1], x[:, 1:]
            G -= 1
        c1, i1 = kw['center1'], kw['center1_idx']
        c2, i2 = kw['center2'], kw['center2_idx']
        G2 = c2.shape[1]
        R = c1.shape[1]
        if kw.get('gather_idx'):
            nb = torch.gather(x, 1, i1.reshape(B, -1, 1).expand(-1, -1, self.dim)).reshape(B * G2, -1, self.dim)
            ctr = torch.gather(x, 1, i2.reshape(B, -1, 1).expand(-1, -1, self.dim)).reshape(B, G2, self.dim)
        else:
            # NB: i1/i2 carry batch offsets b*R (R = 64 centres) but address a (B*G)-row view with
            # G = prompts + 64 rows per sample -- the reference's behaviour (:291-292), reproduced.
            flat = x.reshape(B * G, -1)
            nb = flat[i1, :].reshape(B * G2, -1, self.dim)
            ctr = flat[i2, :].reshape(B, G2, self.dim)
        nb = self._residual(nb, nb)
        ctr = pooling(nb.reshape(B, G2, -1, self.dim), transform=self.bnorm) + 0.3 * ctr
        prompts = x[:, :-R]
        x = propagate(xyz1=c1, xyz2=c2, points1=x[:, -R:], points2=ctr, de_neighbors=8, dist_e=1e-3)
        parts = (cls_x, prompts, x) if is_cls else (prompts, x)
        return torch.cat(parts, dim=1), prompts

    def _residual(self, x, branch):
        """x + drop_path(branch) in one pass (addcmul with the per-sample stochastic-depth factor)."""
        scale = self.drop_path.sample_scale(branch) if isinstance(self.drop_path, DropPath) else None
        return x + branch if scale is None else torch.addcmul(x, branch, scale)

    # -- fused gfx950 path ---------------------------------------------------------------------------
    def _propagate_fused(self, x, kw):
        """_propagate_prompts on the row kernels of csrc/prop.hip.  The neighbour weights and the absolute row
        indices depend only on the centres / the token layout, so they are built once per forward (shared dict
        `_prop_cache` put into the kwargs by TransformerEncoder) with exactly the reference's torch ops."""
        B, Lp, D = x.shape
        off = 1 if kw.get('classification') else 0
        c1, c2 = kw['center1'], kw['center2']
        T, G2 = c1.shape[1], c2.shape[1]
        cache = kw.get('_prop_cache')
        key = (Lp, off)
        pre = kw.get('_prop_entry')        # built ahead of time (pipelined front-end) for exactly this token layout?
        if pre is not None and pre[0] == (B, Lp, off):
            entry = pre[1]
        elif cache is None or key not in cache:
            entry = build_prop_index(c1, c2, kw['center1_idx'], kw['center2_idx'], bool(kw.get('gather_idx')), B, Lp, off)
            if cache is not None:
                cache[key] = entry
        else:
            entry = cache[key]
        w8 = entry.w8
        if not _no_grad_needed(c1, c2):
            # stage 2 of the recipe: the centres carry a gradient back to the prompters, and the interpolation weights 1 / (d + eps)
            # are functions of them -- one autograd node per forward (HF.prop_weights), shared by every block like the index itself
            wkey = ('w8',) + key
            slot = None if cache is None else cache.get(wkey)
            if slot is None:
                w8 = HF.prop_weights(c1, c2, entry, eps=1e-3)
                # every block of the path reads the weights: one alias per consumer, so that their gradients come back summed by ONE
                # launch (HF.fan_out) instead of consumers - 1 element-wise additions of autograd
                n = int(kw.get('_prop_consumers') or 1)
                slot = [w8, list(HF.fan_out(w8, n)) if n > 1 else []]
                if cache is not None:
                    cache[wkey] = slot
            w8 = slot[1].pop() if slot[1] else slot[0]
        u, keep = None, 1.0
        if isinstance(self.drop_path, DropPath) and self.training and self.drop_path.drop_prob > 0:
            u = UNIFORMS.take((B * G2,), x.device)
            keep = 1.0 - self.drop_path.drop_prob
        bn = self.bnorm
        if self.training and bn.track_running_stats:
            bump_counter(bn.num_batches_tracked)
        if bn.affine and (bn.track_running_stats or self.training) and B * Lp <= 15360 and not sync_bn_active(self.training):
            return HF.propagate(x, bn, entry, u, keep, self.training, w8=w8)
        pooled = HF.prop_pool(x, entry.i1, u, keep)
        lc = _bn_rows(pooled, bn, self.training).view(B, G2, D)
        return HF.prop_interp(x, lc, entry.i2, entry.idx8, w8)

    def fusable(self, x):
        return (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] <= 512 and self.attn.fusable(x)
                and isinstance(self.norm1, nn.LayerNorm) and self.mlp.drop.p == 0 and self.attn.proj_drop.p == 0
                and x.shape[1] + 16 <= 144)

    def forward_fused(self, x, pos, **kw):
        """Same function as forward(x + pos, **kw); the element-wise glue runs in the row kernels of
        csrc/block.hip (pos add + prompt insert + norm1 | drop-path residual + norm2 | drop-path residual
        + prompt strip + adapter LayerNorm) and the attention core in attn_fwd/bwd."""
        path = kw['path']
        is_cls = bool(kw.get('classification', False))
        prompts = getattr(self, f'{path}_prompts', None) if path in _PATHS else None
        B, L, D = x.shape
        P = 0 if prompts is None else prompts.shape[0]
        ins = (HF.ROW_INSERT_CLS if is_cls else HF.ROW_INSERT) if P else HF.ROW_IDENTITY
        rem = (HF.ROW_STRIP_CLS if is_cls else HF.ROW_STRIP) if P else HF.ROW_IDENTITY
        u = None
        keep = 1.0
        if isinstance(self.drop_path, DropPath) and self.training and self.drop_path.drop_prob > 0:
            u = UNIFORMS.take((2, B), x.device)
            keep = 1.0 - self.drop_path.drop_prob
        n1, n2 = self.norm1, self.norm2
        attn, mlp = self.attn, self.mlp
        fused_attn = _frozen_bias(attn.proj) and attn.proj_drop.p == 0
        xa, h1 = HF.rowln(x, add=pos, prompts=prompts, mode=ins, P=P, gamma=n1.weight, beta=n1.bias, eps=n1.eps,
                          cls_add=kw.get('_cls_pos'))
        # The four Linear layers run on upp_linear_f32 (csrc/linear.hip).  Frozen output biases ride along in the row kernel
        # that consumes the GEMM result (proj.bias, fc2.bias); fc1's bias, the GELU and -- for backward -- GELU' are the
        # epilogue of the fc1 GEMM, and the fc2 data gradient multiplies by that GELU' in its own epilogue.
        yb = mb = None
        fc1, fc2 = mlp.fc1, mlp.fc2
        fused_mlp = (_frozen_bias(fc1) and _frozen_bias(fc2) and isinstance(mlp.act, nn.GELU) and mlp.act.approximate == 'none'
                     and HF.linear_usable(xa, fc1.weight) and fc1.out_features % 32 == 0 and _no_grad_needed(fc1.weight, fc2.weight))
        u0 = None if u is None else u[0]
        if fused_attn:
            qkv = HF.linear(h1, attn.qkv.weight, attn.qkv.bias)
            ctx = HF.attention(qkv, attn.num_heads, attn.scale)
            y, yb = HF.linear(ctx, attn.proj.weight), attn.proj.bias
        else:
            y = attn(h1)
        x2, h2 = HF.rowln(xa, y=y, ybias=yb, u=u0, keep=keep, gamma=n2.weight, beta=n2.bias, eps=n2.eps)
        if fused_mlp:
            m, mb = HF.mlp_gelu(h2, fc1.weight, fc1.bias, fc2.weight), fc2.bias
        elif _frozen_bias(fc1) and _frozen_bias(fc2) and isinstance(mlp.act, nn.GELU) and fc1.out_features % 4 == 0:
            hid = HF.bias_gelu(HF.linear(h2, fc1.weight), fc1.bias)
            m, mb = HF.linear(hid, fc2.weight), fc2.bias
        else:
            m = mlp(h2)
        adapter = getattr(self, f'{path}_adapter') if (path in _PATHS and kw.get(f'{path}_adapter', False)) else None
        if path in _PATHS and kw.get(f'{path}_adapter', False):
            assert adapter is not None, 'No adapter inserted in block!'
        u2 = None if u is None else u[1]
        if P and kw.get('prompt_propagation_after'):
            x3, _ = HF.rowln(x2, y=m, ybias=mb, u=u2, keep=keep)
            if kw['center1_idx'].numel() == B * kw['center2'].shape[1] * 8 and kw['center1'].dtype == torch.float32 and POOL_TRACE is None:
                x3 = self._propagate_fused(x3, kw)
            else:
                if POOL_TRACE is None:
                    HF.note_declined("Block prompt propagation", "level-2 groups of other than 8 neighbours / centres not f32")
                x3, _ = self._propagate_prompts(x3, kw)
            m, mb, u2, x2 = None, None, None, x3
        if adapter is None:
            x4, _ = HF.rowln(x2, y=m, ybias=mb, u=u2, keep=keep, mode=rem, P=P)
            return x4
        ln = adapter.layer_norm
        if D == 384 and adapter.ln1.weight.shape[0] == 32 and isinstance(adapter.activate, nn.GELU) and FUSE_LN_ADAPTER:
            # one launch: residual + strip + the adapter's LayerNorm + the adapter (csrc/adapter.hip ln_adapter_fwd_kernel)
            pd = adapter.dropout.p if self.training else 0.0
            Lo = x2.shape[1] - (P if rem != HF.ROW_IDENTITY else 0)
            ud = UNIFORMS.take((B * Lo, 32), x.device) if pd > 0 else None
            return HF.ln_adapter(x2, m, mb, u2, keep, rem, P, ln, adapter.ln1.weight, adapter.ln1.bias, adapter.ln2.weight,
                                 adapter.ln2.bias, ud, pd, 0.7)
        x4, ha = HF.rowln(x2, y=m, ybias=mb, u=u2, keep=keep, mode=rem, P=P, gamma=ln.weight, beta=ln.bias, eps=ln.eps)
        if D == 384 and adapter.ln1.weight.shape[0] == 32 and isinstance(adapter.activate, nn.GELU):
            pd = adapter.dropout.p if self.training else 0.0
            ud = UNIFORMS.take((x4.shape[0] * x4.shape[1], 32), x.device) if pd > 0 else None
            return HF.adapter(ha, x4, adapter.ln1.weight, adapter.ln1.bias, adapter.ln2.weight, adapter.ln2.bias, ud, pd, 0.7)
        z = adapter.ln2(adapter.dropout(adapter.activate(adapter.ln1(ha))))
        return torch.add(x4, z, alpha=0.7)

    def forward(self, x, **kw):
        path = kw['path']
        is_cls = bool(kw.get('classification', False))
        prompts = getattr(self, f'{path}_prompts', None) if path in _PATHS else None
        prompt_tokens = None
        if prompts is not None:
            prompt_tokens = prompts.repeat(x.shape[0], 1, 1)
            x = torch.cat((x[:, 0:1], prompt_tokens, x[:, 1:]), 1) if is_cls else torch.cat((prompt_tokens, x), 1)

        x = self._residual(x, self.attn(self.norm1(x)))
        x = self._residual(x, self.mlp(self.norm2(x)))

        if prompt_tokens is not None:
            if kw.get('prompt_propagation_after'):
                x, prompt_tokens = self._propagate_prompts(x, kw)
            t = prompt_tokens.shape[1]
            x = torch.cat((x[:, 0:1], x[:, t + 1:]), 1) if is_cls else x[:, t:]

        if path in _PATHS and kw.get(f'{path}_adapter', False):
            adapter = getattr(self, f'{path}_adapter')
            assert adapter is not None, 'No adapter inserted in block!'
            x = x + adapter(x)
        return x


def _make_blocks(embed_dim, depth, num_heads, mlp_ratio, qkv_bias, qk_scale, drop_rate, attn_drop_rate,
                 drop_path_rate, kwargs):
    return nn.ModuleList([
        Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
              drop=drop_rate, attn_drop=attn_drop_rate,
              drop_path=drop_path_rate[i] if isinstance(drop_path_rate, list) else drop_path_rate,
              block_idx=i, **kwargs)
        for i in range(depth)])


def _run_blocks(blocks, x, pos_i, kwargs, features):
    """The block loop of the encoder / decoder (reference models/Point_MAE_unify.py:288-294, models/Point_MAE_pretask_dev.py:374-377):
    `pos` is added in front of every block."""
    for idx, block in enumerate(blocks):
        x = block.forward_fused(x, pos_i[idx], **kwargs) if block.fusable(x) else block(x + pos_i[idx], **kwargs)
        if features is not None and idx in (3, 7, 11):
            features.append(x)
    return x


class TransformerEncoder(nn.Module):
    """models/Point_MAE_unify.py:273-298: pos is re-added before every block; the pretask /
    rectify paths stop early at their configured depth."""

    def __init__(self, embed_dim=768, depth=4, num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., **kwargs):
        super().__init__()
        self.blocks = _make_blocks(embed_dim, depth, num_heads, mlp_ratio, qkv_bias, qk_scale, drop_rate,
                                   attn_drop_rate, drop_path_rate, kwargs)

    def forward(self, x, pos, **kwargs):
        depth = len(self.blocks)
        if kwargs.get('pretask_depth') and kwargs['path'] == 'pretask':
            depth = kwargs['pretask_depth']
        elif kwargs.get('rectify_depth') and kwargs['path'] == 'rectify':
            depth = kwargs['rectify_depth']
        if 'center1' in kwargs:
            kwargs['_prop_cache'] = {}      # per-forward scratch shared by the blocks (see Block._propagate_fused)
            path = kwargs.get('path')
            kwargs['_prop_consumers'] = (sum(1 for b in self.blocks[:depth] if getattr(b, f'{path}_prompts', None) is not None)
                                         if (path in _PATHS and kwargs.get('prompt_propagation_after')) else 1)
        cls_pos = kwargs.pop('cls_pos_param', None)
        if cls_pos is not None and all(b.fusable(x) for b in self.blocks[:depth]):
            # `pos` is re-added in front of every block, so autograd would sum its (B,L,D) gradient once per block although
            # only the cls row is trainable.  Detach it; every block's row kernel reduces the cls row of its input
            # gradient into cls_pos instead (deferred with the other parameter-gradient sums under TrainStep).
            pos = pos.detach()
            kwargs['_cls_pos'] = cls_pos
        features = []                       # outputs of blocks 3, 7, 11 for the segmentation head
        # `pos` with a gradient (stage 2, pre-training: it comes out of a trainable / differentiated position MLP) is read by every block:
        # hand each block its own alias so that the depth gradients are summed by ONE launch instead of depth - 1 (HF.fan_out)
        pos_i = HF.fan_out(pos, depth)
        x = _run_blocks(self.blocks[:depth], x, pos_i, kwargs, features if kwargs.get('feature_list') else None)
        return features if kwargs.get('feature_list') else x


class TransformerDecoder(nn.Module):
    """models/Point_MAE_pretask_dev.py:352-384"""

    def __init__(self, embed_dim=384, depth=4, num_heads=6, mlp_ratio=4., qkv_bias=False, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm, **kwargs):
        super().__init__()
        self.blocks = _make_blocks(embed_dim, depth, num_heads, mlp_ratio, qkv_bias, qk_scale, drop_rate,
                                   attn_drop_rate, drop_path_rate, kwargs)
        self.norm = norm_layer(embed_dim)
        self.head = nn.Identity()
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, x, pos, return_token_num, **kwargs):
        x = _run_blocks(self.blocks, x, HF.fan_out(pos, len(self.blocks)), kwargs, None)
        return self.head(HF.layer_norm(x, self.norm, last=return_token_num))      # (strip row map: no copy of the row window)


# --------------------------------------------------------------------------- rectify prompter
class PositionalEmbedding(nn.Module):
    """x -> (x, sin(2^k x), cos(2^k x))_k  (models/Point_MAE_pretask_dev.py:22-52)."""

    def __init__(self, N_freqs, logscale=True):
        super().__init__()
        self.N_freqs = N_freqs
        self.freq_bands = (2 ** torch.linspace(0, N_freqs - 1, N_freqs) if logscale
                           else torch.linspace(1, 2 ** (N_freqs - 1), N_freqs)).tolist()

    def forward(self, x, out=None, col0=0):
        if x.is_cuda and x.dtype == torch.float32 and len(self.freq_bands) <= 8 and _no_grad_needed(x):
            return HF.posenc(x, self.freq_bands, out, col0)    # one launch; may write a column window of `out`
        assert out is None
        out = [x]
        for freq in self.freq_bands:
            out += [torch.sin(freq * x), torch.cos(freq * x)]
        return torch.cat(out, -1)


class PointNetSetAbstraction(nn.Module):
    """models/Point_MAE_pretask_dev.py:386-423"""

    def __init__(self, num_group, group_size, in_channel, mlp):
        super().__init__()
        self.group_divider = Group(num_group, group_size)
        self.num_group = num_group
        self.group_size = group_size
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last = out_channel

    def forward(self, xyz, points):
        B, N, _ = xyz.shape
        if points.is_cuda and points.dtype == torch.float32 and POOL_TRACE is None:
            # per-sample indices and a gather whose backward is the deterministic pull kernel (torch's indexing backward sorts on the device)
            _, center, idx, _ = self.group_divider(xyz.float(), require_index=True, gather_idx=True)
            x = HF.gather_rows(points.reshape(B, N, -1), idx.reshape(B, -1)).reshape(B * self.num_group * self.group_size, -1)
        else:
            _, center, idx, _ = self.group_divider(xyz if xyz.dtype == torch.float64 else xyz.float(), require_index=True)   # (f64: arbitration runs of the tests)
            x = points.reshape(B * N, -1)[idx]                          # (B*G*k, C) rows
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            x = _pointwise_bn_relu(x, conv, bn, self.training)
        x = max_over(x.view(B, self.num_group, self.group_size, -1), 2, 'set_abstraction')
        return center.reshape(B, self.num_group, -1), x


class PointNetFeaturePropagation(nn.Module):
    """models/Point_MAE_pretask_dev.py:425-473"""

    def __init__(self, in_channel, mlp, interpolate_neighbors=16):
        super().__init__()
        self.interpolate_neighbors = interpolate_neighbors
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last = out_channel

    def forward(self, xyz1, xyz2, points1, points2, cat_buffer=None):
        """cat_buffer: optional (B,N,C1+C2) buffer whose first C1 columns already hold points1 (fused path)."""
        N = xyz1.shape[1]
        if cat_buffer is not None:
            _inverse_distance_interp(xyz1, xyz2, points2, self.interpolate_neighbors, 1e-4, out=cat_buffer,
                                     col0=cat_buffer.shape[-1] - points2.shape[-1])
            x = cat_buffer
        elif self.commute_first_layer and 1 < xyz2.shape[1] and 4 * xyz2.shape[1] <= N and points2.is_cuda:
            return self._forward_commuted(xyz1, xyz2, points1, points2)
        else:
            if xyz2.shape[1] == 1:
                interp = points2.repeat(1, N, 1)
            else:
                interp = _inverse_distance_interp(xyz1, xyz2, points2, self.interpolate_neighbors, 1e-4)
            x = interp if points1 is None else torch.cat([points1, interp], dim=-1)
        B = x.shape[0]
        x = x.reshape(B * N, -1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            x = _pointwise_bn_relu(x, conv, bn, self.training)
        return x.view(B, N, -1)

    # The first 1x1 convolution and the interpolation commute: the interpolated feature of a point is a convex combination
    # (weights sum to 1) of the S source rows, so  W.[p1 | sum_k w_k x_k] + b  =  W_p.p1 + sum_k w_k (W_x.x_k + b).
    # The (B N)-row GEMM of the reference formulation (65,536 x 1155 x 1536 for the part-segmentation head: a third of the
    # step's FLOPs, forward and both gradients) becomes a (B S)-row one -- 16x fewer rows at S = 128, N = 2048 -- plus an
    # interpolation of C_out- instead of C_in-wide rows and a K = C1 (3) rank update.  Same function; f32 rounding differs
    # at the 1e-7 level (the sum over k and the product with W are re-associated).
    commute_first_layer = True

    def _forward_commuted(self, xyz1, xyz2, points1, points2):
        B, N = xyz1.shape[0], xyz1.shape[1]
        conv, bn = self.mlp_convs[0], self.mlp_bns[0]
        w = conv.weight.view(conv.weight.shape[0], -1)
        C1 = 0 if points1 is None else points1.shape[-1]
        z = HF.linear(points2, w[:, C1:], conv.bias)                                        # (B,S,C_out)
        S, k, Co = xyz2.shape[1], self.interpolate_neighbors, w.shape[0]
        if (C1 == 3 and z.dtype == torch.float32 and S <= 256 and k <= min(4, S) and Co >= 256 and Co % 4 == 0 and N <= 4096
                and xyz1.shape[-1] == 3 and _no_grad_needed(xyz1, xyz2, points1) and POOL_TRACE is None):
            # one launch: neighbour search; one launch: interpolation of the C_out-wide rows + the rank-3 xyz term
            dists, idx = HF.sqdist_topk(xyz1, xyz2, k)
            y = HF.interp_affine_train(dists, idx, z, points1, w[:, :C1].t(), k, 1e-4).reshape(B * N, -1)
        else:
            y = _inverse_distance_interp(xyz1, xyz2, z, k, 1e-4).reshape(B * N, -1)
            if C1:
                # (the concat's first C1 columns as a separate small-K product: forward, data and weight gradient on our kernels)
                y = y + HF.linear(points1.reshape(B * N, C1), w[:, :C1]) if y.is_cuda else torch.addmm(y, points1.reshape(B * N, C1), w[:, :C1].t())
        if self.training and bn.track_running_stats:
            bump_counter(bn.num_batches_tracked)
        x = _bn_rows(y, bn, self.training, relu=True)
        for conv, bn in zip(self.mlp_convs[1:], self.mlp_bns[1:]):
            x = _pointwise_bn_relu(x, conv, bn, self.training)
        return x.view(B, N, -1)


class RectifyPrompter(nn.Module):
    """Per-point rectification vector from the rectify-path tokens
    (models/Point_MAE_pretask_dev.py:475-517)."""

    def __init__(self, in_channels, out_channels, hidden_dimesion=384, embedding_level=4, num_group=32,
                 group_size=16, top_center_dim=12):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.hidden_dimesion = hidden_dimesion
        self.num_group = num_group
        self.group_size = group_size
        self.top_center_dim = top_center_dim
        self.position_embedding = PositionalEmbedding(embedding_level)
        self.abstraction = PointNetSetAbstraction(num_group, group_size, hidden_dimesion, mlp=[64, 32, top_center_dim])
        self.propagation1 = PointNetFeaturePropagation(in_channel=in_channels * (2 * embedding_level + 1) + 32, mlp=[32, 32])
        self.propagation2 = PointNetFeaturePropagation(in_channel=top_center_dim, mlp=[64, 32])
        self.score_head = nn.Sequential(nn.Linear(32, 64), nn.ReLU(), nn.Dropout(0.2), nn.Linear(64, out_channels))
        self.score_factor = 1.0
        for layer in self.score_head:
            if isinstance(layer, nn.Linear):
                nn.init.kaiming_uniform_(layer.weight, a=math.sqrt(5.0))
                nn.init.constant_(layer.bias, val=0.0)

    def forward(self, x, center1, center1_feature, require_shape_feature=False):
        feature, shape_feature = self.features(x, center1, center1_feature)
        l0, _, drop, l1 = self.score_head                                  # Linear(32, 64), ReLU, Dropout(0.2), Linear(64, 3): both on our kernels
        noise_score = HF.linear(drop(_relu_linear(feature, l0)), l1.weight, l1.bias) * self.score_factor
        return (noise_score, shape_feature) if require_shape_feature else noise_score

    def select(self, x, center1, center1_feature, keep, nudge=0.2):
        """The model's use of the prompter in one call (reference models/Point_MAE_unify.py:553-559): score every point, move the cloud by
        nudge * predicted offset, return the `keep` least suspicious points in descending-score order.  Two launches behind the
        per-point feature when no gradient is asked for (upp_rectify_select); the reference formulation otherwise."""
        feature, _ = self.features(x, center1, center1_feature)
        l0, _, drop, l1 = self.score_head
        if (feature.is_cuda and feature.dtype == torch.float32 and x.shape[1] <= 16384 and 0 < keep <= x.shape[1] and tuple(l0.weight.shape) == (64, 32)
                and tuple(l1.weight.shape) == (3, 64) and l0.bias is not None and l1.bias is not None
                and _no_grad_needed(feature, x, l0.weight, l0.bias, l1.weight, l1.bias) and POOL_TRACE is None):
            from upp_hip import ops
            live = self.training and drop.p > 0
            u = UNIFORMS.take((x.shape[0] * x.shape[1], 64), x.device) if live else None
            return ops.rectify_select(feature.contiguous(), l0.weight, l0.bias, l1.weight, l1.bias, x.contiguous(), keep, u,
                                      drop.p if live else 0.0, self.score_factor, nudge)
        pred = HF.linear(drop(_relu_linear(feature, l0)), l1.weight, l1.bias)
        if self.score_factor != 1.0:                   # (1.0 in every shipped configuration: a multiplication and its backward saved)
            pred = pred * self.score_factor
        with torch.no_grad():
            score = torch.norm(pred, p=2, dim=-1)
        order = trace_idx('rectify.order', HF.argsort_rows(score, descending=True))
        moved = x + pred * nudge           # (NOT torch.add(x, pred, alpha=nudge): the device fuses that into one rounding, the host does not, and the
                                           #  ill-conditioned interpolation weights downstream turn an ulp of the coordinates into 2.5e-4 of mask_token's gradient)
        # (HF.gather_rows: forward and a sort-free, deterministic backward on the interpolation kernels instead of torch's gather +
        #  zero-fill + scatter_add)
        return HF.gather_rows(moved, order[:, -keep:].contiguous())

    def features(self, x, center1, center1_feature):
        """-> (per-point feature (B,N,32) in front of the score head, shape feature (B, num_group * top_center_dim))."""
        B = center1_feature.shape[0]
        center2, center2_feature = self.abstraction(center1, center1_feature)
        shape_feature = center2_feature.reshape(B, -1)
        center1_feature = self.propagation2(center1, center2, None, center2_feature)
        pe = self.position_embedding
        if (x.is_cuda and x.dtype == torch.float32 and len(pe.freq_bands) <= 8 and center1.shape[1] > 1
                and self.propagation1.interpolate_neighbors <= 16 and _no_grad_needed(x, center1, center1_feature) and POOL_TRACE is None):
            # embedding and interpolation write the two halves of one buffer (the reference's torch.cat)
            C1, C2 = 3 * (2 * len(pe.freq_bands) + 1), center1_feature.shape[-1]
            buf = x.new_empty(x.shape[0], x.shape[1], C1 + C2)
            pe(x, out=buf, col0=0)
            feature = self.propagation1(x, center1, None, center1_feature, cat_buffer=buf)
        else:
            feature = self.propagation1(x, center1, pe(x), center1_feature)
        return feature, shape_feature
