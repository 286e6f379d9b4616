"""Autograd Functions over upp_hip.ops with the reference operators' contracts."""
import os
import sys
import weakref

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import ops


class FurthestPointSampling(Function):
    """pointnet2_utils.furthest_point_sample: (B,N,3) f32, npoint -> (B,npoint) int32, non-differentiable."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        idx = ops.fps(xyz, npoint)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad_out):
        return None, None


class GatherOperation(Function):
    """pointnet2_utils.gather_operation: features (B,C,N), idx (B,M) int32 -> (B,C,M); grad w.r.t. features."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.N = features.shape[2]
        return ops.gather_fwd(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return ops.gather_bwd(grad_out.contiguous(), idx, ctx.N), None


class _FpsGather(Function):
    """FPS + coordinate gather in one launch (reference utils/misc.py:13-20 fps())."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        idx, centers = ops.fps(xyz, npoint, want_centers=True)
        ctx.save_for_backward(idx)
        ctx.N = xyz.shape[1]
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)          # (no zero tensor for the index output's "gradient": a fill launch per backward)
        return centers, idx

    @staticmethod
    def backward(ctx, grad_centers, _grad_idx):
        if grad_centers is None:
            return None, None
        (idx,) = ctx.saved_tensors
        return ops.fps_gather_bwd(grad_centers.contiguous(), idx, ctx.N), None        # (one launch: zero-fill + int32 indices + scatter)


class _KnnGroup(Function):
    """kNN + neighbourhood gather + centre subtraction in one launch
    (reference models/Point_MAE_unify.py:69-88).  Differentiable w.r.t. xyz and center
    through the gather/subtraction, not through the neighbour selection."""

    @staticmethod
    def forward(ctx, xyz, center, k):
        _, idx, neigh = ops.knn(xyz, center, k, want_dist=False, want_neigh=True)
        ctx.save_for_backward(idx)
        ctx.N = xyz.shape[1]
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return neigh, idx

    @staticmethod
    def backward(ctx, grad_neigh, _grad_idx):
        if grad_neigh is None:
            return None, None, None
        (idx,) = ctx.saved_tensors
        gx, gc = ops.group_bwd(grad_neigh.contiguous(), idx, ctx.N,
                               need_xyz=ctx.needs_input_grad[0], need_center=ctx.needs_input_grad[1])
        return gx, gc, None


class _GroupPoints(Function):
    """out[b,g,k] = xyz[b, idx[b,g,k]] - center[b,g] for given indices."""

    @staticmethod
    def forward(ctx, xyz, center, idx):
        ctx.save_for_backward(idx)
        ctx.N = xyz.shape[1]
        return ops.group_fwd(xyz, center, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        gx, gc = ops.group_bwd(grad_out.contiguous(), idx, ctx.N,
                               need_xyz=ctx.needs_input_grad[0], need_center=ctx.needs_input_grad[1])
        return gx, gc, None


class ChamferFunction(Function):
    """extensions.chamfer_dist.ChamferFunction (reference extensions/chamfer_dist/__init__.py:13-25)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        dist1, dist2, idx1, idx2 = ops.chamfer_fwd(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, grad_dist1, grad_dist2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        return ops.chamfer_bwd(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2)


class _ChamferLoss(Function):
    """ChamferDistanceL1 / L2 as one node: two direction kernels + the fused reduction forward (4 launches), a scalar x factor pair and the
    gradient kernel backward -- torch's formulation issues ~7 launches forward and ~8 backward around the same two kernels."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, l1):
        dist1, dist2, idx1, idx2 = ops.chamfer_fwd(xyz1, xyz2)
        loss, fac1, fac2 = ops.chamfer_loss(dist1, dist2, l1)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2, fac1, fac2)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        xyz1, xyz2, idx1, idx2, fac1, fac2 = ctx.saved_tensors
        g1, g2 = ops.chamfer_bwd(xyz1, xyz2, idx1, idx2, fac1 * g, fac2 * g)
        return g1, g2, None


def chamfer_loss(xyz1, xyz2, l1=True):
    """ChamferDistanceL1 (l1) / ChamferDistanceL2 of two batches of clouds (reference extensions/chamfer_dist/__init__.py:44-84)."""
    return _ChamferLoss.apply(xyz1.contiguous(), xyz2.contiguous(), bool(l1))


class EarthMoverDistanceFunction(Function):
    """extensions.emd.emd.EarthMoverDistanceFunction (reference extensions/emd/emd.py:5-21)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1 = xyz1.contiguous()
        xyz2 = xyz2.contiguous()
        assert xyz1.is_cuda and xyz2.is_cuda, "Only support cuda currently."
        match = ops.emd_approxmatch(xyz1, xyz2)
        cost = ops.emd_matchcost(xyz1, xyz2, match)
        ctx.save_for_backward(xyz1, xyz2, match)
        return cost

    @staticmethod
    def backward(ctx, grad_cost):
        xyz1, xyz2, match = ctx.saved_tensors
        return ops.emd_matchcost_bwd(grad_cost.contiguous(), xyz1, xyz2, match)


furthest_point_sample = FurthestPointSampling.apply
gather_operation = GatherOperation.apply


def _torch_cpu(*tensors):
    """CPU tensors and the opt-in torch formulations switched on (upp_hip.torch_cpu: BASELINE configs[0], plumbing on a GPU-less host)?
    Otherwise the operators below go to upp_hip.ops, which serves HIP tensors only and says so."""
    from . import torch_cpu
    return torch_cpu.enabled() and all(isinstance(t, torch.Tensor) and not t.is_cuda for t in tensors)


def fps_gather(xyz, npoint):
    """-> (centers (B,npoint,3), idx (B,npoint) int32)."""
    if _torch_cpu(xyz):
        from . import torch_cpu
        return torch_cpu.fps(xyz, int(npoint))
    return _FpsGather.apply(xyz, int(npoint))


def knn_query(ref, query, k):
    """-> (dist (B,Q,k) f32, idx (B,Q,k) int64); no gradient (the reference wraps it in no_grad)."""
    if _torch_cpu(ref, query):
        from . import torch_cpu
        return torch_cpu.knn(ref, query, int(k))
    with torch.no_grad():
        dist, idx, _ = ops.knn(ref, query, k, want_dist=True, want_neigh=False)
    return dist, idx


def knn_group(xyz, center, k):
    """-> (neighborhood (B,G,k,3) centred on `center`, idx (B,G,k) int64)."""
    if _torch_cpu(xyz, center):
        from . import torch_cpu
        return torch_cpu.knn_group(xyz, center, int(k))
    return _KnnGroup.apply(xyz, center, int(k))


def group_points(xyz, center, idx):
    return _GroupPoints.apply(xyz, center, idx)


# ------------------------------------------------------------------ Transformer block glue
ROW_IDENTITY, ROW_INSERT_CLS, ROW_INSERT, ROW_STRIP_CLS, ROW_STRIP = 0, 1, 2, 3, 4


class _RowLN(Function):
    """rows = map(x (+add), prompts; mode, P) (+ dp_scale(u) * y);  returns (rows, LayerNorm(rows)).
    One kernel forward, one backward (+ one for trainable LayerNorm parameters).  See upp_rowln_fwd."""

    @staticmethod
    def forward(ctx, x, add, prompts, y, gamma, beta, mode, P, u, keep, eps, cls_add=None, ybias=None):
        x = x.contiguous()
        B, Lin, D = x.shape
        Lout = Lin + P if mode in (ROW_INSERT_CLS, ROW_INSERT) else (Lin - P if mode in (ROW_STRIP_CLS, ROW_STRIP) else Lin)
        add_c = add.contiguous() if add is not None else None
        y_c = y.contiguous() if y is not None else None
        xo, h, mean, rstd = ops.rowln_fwd(x, add_c, prompts, mode, P, y_c, u, keep, gamma, beta, eps, Lout, ybias=ybias)
        ctx.save_for_backward(xo, mean, rstd, gamma, u)
        ctx.dims = (B, Lin, Lout, D, P, mode)
        ctx.keep = keep
        ctx.has = (add is not None, prompts is not None, y is not None, gamma is not None)
        ctx.param_ptrs = tuple(t.data_ptr() if t is not None else 0 for t in (prompts, gamma, beta, cls_add))
        ctx.cls_shape = tuple(cls_add.shape) if cls_add is not None else None
        if gamma is None:
            return xo, xo.new_empty(0)
        return xo, h

    @staticmethod
    def backward(ctx, g_xo, g_h):
        xo, mean, rstd, gamma, u = ctx.saved_tensors
        B, Lin, Lout, D, P, mode = ctx.dims
        has_add, has_prompts, has_y, has_ln = ctx.has
        need = ctx.needs_input_grad
        g_xo = g_xo.contiguous() if g_xo is not None else None
        g_hc = g_h.contiguous() if (has_ln and g_h is not None) else None
        if g_xo is None and g_hc is None:
            return (None,) * 13
        strip = mode in (ROW_STRIP_CLS, ROW_STRIP) and P > 0
        need_cls = ctx.cls_shape is not None and len(need) > 11 and need[11]
        need_ln = has_ln and g_hc is not None and (need[4] or need[5])
        g_x, g_p, g_y, ln_part = ops.rowln_bwd(g_xo, g_hc, xo, mean, rstd, gamma, mode, u, ctx.keep, B, Lin, Lout, D, P,
                                               need_x=need[0] or (has_add and need[1]) or need_cls,
                                               need_prompt=has_prompts and need[2], need_y=has_y and need[3], need_ln_part=need_ln)
        g_gamma = g_beta = g_prompts = None
        p_prompts, p_gamma, p_beta, p_cls = ctx.param_ptrs
        g_cls = None
        if need_cls:     # `add` holds a trainable row 0 (the cls position) whose gradient is the batch sum of g_x[:, 0]
            _, g_cls = _DEFERRED.reduce(p_cls, g_x.view(B, Lin * D), 0, D)
            if g_cls is not None:
                g_cls = g_cls.view(ctx.cls_shape)
        if need_ln:      # per-workgroup partials [d_gamma | d_beta] written by the same backward pass
            _, g_gamma = _DEFERRED.reduce(p_gamma, ln_part, 0, D)
            _, g_beta = _DEFERRED.reduce(p_beta, ln_part, D, D)
        if g_p is not None:
            _, g_prompts = _DEFERRED.reduce(p_prompts, g_p.view(B, P * D), 0, P * D)
            if g_prompts is not None:
                g_prompts = g_prompts.view(P, D)
        return (g_x if need[0] else None, g_x if (has_add and need[1]) else None, g_prompts, g_y,
                g_gamma if need[4] else None, g_beta if need[5] else None, None, None, None, None, None, g_cls, None)


def rowln(x, add=None, prompts=None, y=None, gamma=None, beta=None, mode=ROW_IDENTITY, P=0, u=None, keep=1.0, eps=1e-5, cls_add=None,
          ybias=None):
    """-> (rows (B,Lout,D), LayerNorm(rows) or None).  mode / P: see ROW_* and upp_rowln_fwd.
    ybias: optional (D) frozen bias added to y (the Linear that produced y then runs its GEMM bias-free).
    cls_add: optional (1,1,D) parameter that `add` (passed detached) carries in its row 0 for every sample; its gradient
    (batch sum of the row-0 input gradient) is produced here instead of through a (B,L,D) gradient of `add`."""
    if cls_add is not None and (add is None or add.requires_grad or mode not in (ROW_IDENTITY, ROW_INSERT_CLS)):
        raise ValueError("cls_add needs a detached `add` and a row map that keeps source row 0 in place")
    if ybias is not None and (y is None or (torch.is_grad_enabled() and ybias.requires_grad)):
        raise ValueError("ybias is the frozen bias of the Linear that produced y")
    xo, h = _RowLN.apply(x, add, prompts, y, gamma, beta, int(mode), int(P), u, float(keep), float(eps), cls_add, ybias)
    return xo, (h if gamma is not None else None)


def layer_norm(x, ln, last=None):
    """ln(x) -- or, with `last`, ln(x[:, -last:]) -- for an nn.LayerNorm over the last dimension of a (B,L,D) / (R,D) f32 HIP tensor on
    the row kernel of csrc/block.hip (upp_rowln_fwd: identity / strip row map, the normalised rows only); the module itself for anything
    else.  The prompting front-end calls it (reference models/Point_MAE_unify.py:588 `self.norm`, models/Point_MAE_pretask_dev.py:381
    the decoder's norm over the returned tokens): no torch kernel, no copy of the row window."""
    ok = (isinstance(ln, torch.nn.LayerNorm) and x.is_cuda and x.dtype == torch.float32 and x.dim() in (2, 3) and len(ln.normalized_shape) == 1
          and ln.normalized_shape[0] == x.shape[-1] and x.shape[-1] <= 512 and ln.elementwise_affine and ln.bias is not None and x.numel() > 0)
    if not ok:
        return ln(x if last is None else x[:, -last:])
    x3 = x if x.dim() == 3 else x.unsqueeze(0)
    L = x3.shape[1]
    strip = 0 if last is None else L - int(last)
    if strip < 0:
        raise ValueError("layer_norm: last = %d of %d rows" % (last, L))
    mode = ROW_STRIP if strip else ROW_IDENTITY
    if torch.is_grad_enabled() and (x.requires_grad or ln.weight.requires_grad or ln.bias.requires_grad):
        _, h = rowln(x3, gamma=ln.weight, beta=ln.bias, mode=mode, P=strip, eps=ln.eps)
    else:
        _, h, _, _ = ops.rowln_fwd(x3.contiguous(), None, None, mode, strip, None, None, 1.0, ln.weight, ln.bias, float(ln.eps), L - strip, want_xo=False)
    return h if x.dim() == 3 else h.squeeze(0)


class _Attention(Function):
    """ctx = softmax(q k^T * scale) v from the packed qkv projection (B, L, 3*H*64)."""

    @staticmethod
    def forward(ctx, qkv, num_heads, scale):
        qkv = qkv.contiguous()
        B, L, _ = qkv.shape
        out, lse = ops.attn_fwd(qkv, B, L, num_heads, scale)
        ctx.save_for_backward(qkv, out, lse)
        ctx.meta = (B, L, num_heads, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, out, lse = ctx.saved_tensors
        B, L, H, scale = ctx.meta
        return ops.attn_bwd(qkv, out, g.contiguous(), lse, B, L, H, scale), None, None


def attention(qkv, num_heads, scale):
    """softmax(q k^T scale) v per head on the FP32-MFMA attention kernels (L <= 96: attn_flash16.hip, L <= 160: attn_long.hip)."""
    return _Attention.apply(qkv, int(num_heads), float(scale))


# ------------------------------------------------------------------ deferred parameter-gradient sums
class _DeferredSums:
    """Parameter gradients that are sums of partial results (adapter weights, LayerNorm gamma/beta, prompts) are not read
    by anything inside a backward pass.  Inside `with deferred_sums(targets):` (TrainStep) their reductions are queued
    and run as ONE upp_batched_sum launch when the scope exits, ACCUMULATING straight into the gradient buffers the
    caller registered (targets: {parameter data_ptr: zero-initialised f32 buffer of the parameter's shape}); autograd is
    handed None for those parameters, so nothing depends on how it would have stored a returned tensor.  Parameters
    without a registered buffer, and everything outside a scope, are summed at once and returned as usual."""

    TALL = 4096

    def __init__(self):
        self.targets = None
        self.jobs = []
        self.wgrads = []
        self.adapters = []
        self.routed = set()

    def reduce(self, ptr, part, offset, length):
        """Sum columns [offset, offset+length) of the 2-D partial matrix `part` over its rows.
        -> (routed, tensor-or-None): routed sums appear in the registered buffer at scope exit."""
        if part.is_cuda and part.shape[0] > self.TALL and part.dtype == torch.float32 and part.stride(1) == 1:
            # one very tall matrix (a bias gradient over 65,536 point rows): first stage on many workgroups (a single column
            # sum of 64-column workgroups took 0.78 ms of the pre-training step)
            part, offset = ops.colsum_partials(part, offset, length), 0
        dst = self.targets.get(ptr) if (self.targets is not None and part.is_cuda) else None
        if dst is None or dst.numel() != length:
            if part.is_cuda and part.dtype == torch.float32 and part.dim() == 2 and part.stride(1) == 1:
                return False, ops.sum_rows(part, offset, length)                  # (never a torch reduction inside a step: ops.sum_rows)
            return False, part[:, offset:offset + length].sum(dim=0)
        self.jobs.append((part, offset, part.shape[0], length, part.stride(0), dst, True))
        self.routed.add(ptr)
        return True, None

    def wgrad(self, ptr, g2, x2):
        """Queue dW = g2^T . x2 for a parameter whose gradient buffer is registered: every weight gradient of the backward pass
        runs in ONE grouped launch at scope exit (upp_linear_wgrad_grouped_f32; a weight gradient is read by nothing inside the
        pass) and its partials are added to the buffer by the batched sum.  -> True when queued (autograd is handed None)."""
        dst = self.targets.get(ptr) if (self.targets is not None and g2.is_cuda) else None
        if dst is None or dst.numel() != g2.shape[1] * x2.shape[1]:
            return False
        self.wgrads.append((g2, x2, dst))
        self.routed.add(ptr)
        return True

    def wgrad_rows(self, ptr, g2, x2, n_rows):
        """dW = (g2^T . x2)[:n_rows] for a parameter whose output was computed wider than it is (g2 has n_pad >= n_rows columns whose
        pad columns are zero): the partial tiles are (n_pad, K); only their first n_rows rows are summed into the buffer."""
        dst = self.targets.get(ptr) if (self.targets is not None and g2.is_cuda) else None
        if dst is None or dst.numel() != n_rows * x2.shape[1] or n_rows > g2.shape[1]:
            return False
        self.wgrads.append((g2, x2, dst))
        self.routed.add(ptr)
        return True

    def wgrad_window(self, w, g2, x2):
        """The same for a weight that is a COLUMN RANGE w = P[:, a:b] of a registered 2-D parameter P (the segmentation head's first layer
        multiplies [per-point | per-sample] column ranges of one weight; the feature propagation's first layer [xyz | features]): the
        partials are added into that range of P's gradient buffer (upp_batched_sum window destination) -- autograd is handed None, so the
        slice's backward (zero-fill of P's shape, strided copy, add) never runs.  -> True when queued."""
        root = w._base
        if self.targets is None or root is None or not g2.is_cuda or w.dim() != 2 or w.stride(1) != 1:
            return False
        dst = self.targets.get(root.data_ptr())
        N, K = w.shape
        if dst is None or dst.numel() != root.numel() or root.numel() % N or not root.is_contiguous():
            return False
        Kt = root.numel() // N                        # P viewed as (N, Kt): trailing singleton dimensions of a Conv1d weight fold away
        off = w.storage_offset() - root.storage_offset()
        if w.stride(0) != Kt or off < 0 or off + K > Kt or (g2.shape[1], x2.shape[1]) != (N, K):
            return False
        self.wgrads.append((g2, x2, dst.view(N, Kt)[:, off:off + K]))
        self.routed.add(root.data_ptr())
        return True

    def adapter(self, ptrs, sizes, job):
        """Queue the weight gradients of one block's adapter (ptrs / sizes of W1, b1, W2, b2; job: see ops.adapter_wgrad_batched): formed
        from the per-row factors for every block of the backward pass in ONE launch at scope exit, instead of one partial matrix per
        16-row workgroup (153 MB per headline step through upp_batched_sum).  -> True when all four buffers are registered."""
        if self.targets is None:
            return False
        dsts = [self.targets.get(p_) for p_ in ptrs]
        if any(d is None or d.numel() != n for d, n in zip(dsts, sizes)):
            return False
        self.adapters.append((job, dsts))
        self.routed.update(ptrs)
        return True

    def flush(self):
        jobs, self.jobs = self.jobs, []
        wg, self.wgrads = self.wgrads, []
        ad, self.adapters = self.adapters, []
        if ad:
            for part, (_, dsts) in zip(ops.adapter_wgrad_batched([j for j, _ in ad]), ad):
                off = 0
                for dst in dsts:                                   # [dW1 | dW2 | db1 | db2], the order of `ptrs`: W1, W2, b1, b2
                    jobs.append((part, off, part.shape[0], dst.numel(), part.stride(0), dst, True))
                    off += dst.numel()
        if wg:
            for part, (_, _, dst) in zip(ops.linear_wgrad_grouped([(g, x) for g, x, _ in wg]), wg):
                n, full = dst.numel(), part[0].numel()                # (full > n: the pad rows of a wgrad_rows problem are not summed)
                jobs.append((part.view(part.shape[0], full), 0, part.shape[0], n, full, dst, True))
        ops.batched_sum(jobs)


_DEFERRED = _DeferredSums()


class deferred_sums:
    """Scope of one backward pass whose parameter-gradient partial sums are batched (see _DeferredSums)."""

    def __init__(self, targets):
        self.targets = targets
        self.routed = set()      # after exit: data_ptrs of the parameters whose buffers received sums

    def __enter__(self):
        if _DEFERRED.targets is not None:
            raise RuntimeError("deferred_sums scopes do not nest")
        _DEFERRED.targets = self.targets
        _DEFERRED.routed = set()
        return self

    def __exit__(self, *exc):
        _DEFERRED.targets = None
        self.routed = _DEFERRED.routed
        if exc[0] is None:
            _DEFERRED.flush()
        else:
            _DEFERRED.jobs = []
            _DEFERRED.wgrads = []
            _DEFERRED.adapters = []
        return False


# ------------------------------------------------------------------ FP32-MFMA Linear (upp_linear_f32)
_declined = set()


def note_declined(what, why):
    """A fused gfx950 path was not taken: say so ONCE per (site, reason) when UPP_VERBOSE is set -- the torch / library
    path that runs instead is correct but slower, and nothing else would show it."""
    key = (what, why)
    if key in _declined:
        return
    _declined.add(key)
    if os.environ.get("UPP_VERBOSE"):
        print("[upp_hip] %s: fused path declined (%s); running the torch / library formulation" % (what, why), file=sys.stderr)


class _TransposedWeights:
    """W^T copies of FROZEN weights for the data-gradient GEMMs (dX = dY . W is upp_linear_f32(dY, W^T)).  One copy per
    weight, made on first use (during the eager warm-up of a captured step) and REFRESHED IN PLACE when the weight's
    version counter moved (load_state_dict) -- in place, so that HIP graphs that captured the copy's address stay valid;
    `refresh()` re-copies every entry (call it after loading weights into a model whose step is already captured)."""

    def __init__(self):
        self.entries = {}
        self.trainable = {}

    @staticmethod
    def _view_of(owner, geom):
        return torch.as_strided(owner.detach(), geom[0], geom[1], geom[2])

    def get(self, w):
        owner = w._base if w._base is not None else w        # a column window of a parameter (w3[:, 256:]) shares its version counter
        geom = (tuple(w.shape), tuple(w.stride()), w.storage_offset())
        key = (w.data_ptr(),) + geom
        e = self.entries.get(key)
        if e is None or e[0]() is not owner:         # (another tensor on the same address is another weight)
            if len(self.entries) > 1024:
                self.entries = {k: v for k, v in self.entries.items() if v[0]() is not None}
            K = w.shape[1]
            full = torch.zeros(((K + 3) // 4 * 4, w.shape[0]), dtype=w.dtype, device=w.device)   # rows padded to a multiple of 4: get_padded()
            full._upp_persistent = True            # (ops.PLANES may keep the bf16 plane image of this copy: it lives as long as the entry)
            wt = full[:K]
            wt.copy_(w.detach().t())
            self.entries[key] = [weakref.ref(owner), owner._version, wt, geom, full]
            return wt
        if e[1] != owner._version:
            if w.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("upp_hip: a frozen weight changed after its W^T copy was made and the stream is capturing; call "
                                   "upp_hip.functional.refresh_caches(model) after changing weights and before capturing a step")
            e[2].copy_(w.detach().t())
            e[1] = owner._version
        return e[2]

    def get_padded(self, w):
        """W^T with zero rows up to a multiple of 4 (K, ceil4(K) x N): the B operand of a data-gradient GEMM whose output width K is
        not a multiple of 4 (K = 3: the first layer of a position MLP whose input carries a gradient); same storage as get(w)."""
        self.get(w)
        return self.entries[(w.data_ptr(), tuple(w.shape), tuple(w.stride()), w.storage_offset())][4]

    def refresh(self):
        for e in self.entries.values():
            owner = e[0]()
            if owner is not None:
                e[2].copy_(self._view_of(owner, e[3]).t())
                e[1] = owner._version

    # -- trainable weights inside a step driver (TrainStep): their W^T copies are persistent too and are refreshed by ONE batched
    #    launch at the start of every step (`refresh_trainable`, called by TrainStep._forward_backward while `managed` is set) -- the
    #    pre-training recipe transposed 71 weights per step one by one (0.36 ms).  Outside a step driver a trainable weight is
    #    transposed afresh at every use (it may have changed by any means).
    managed = False

    def get_trainable(self, w):
        key = ("t", w.data_ptr(), tuple(w.shape), tuple(w.stride()))
        e = self.trainable.get(key)
        if e is None or e[0]() is None:
            wt = ops.transpose(w.detach())
            self.trainable[key] = [weakref.ref(w), wt, w.is_contiguous()]
            return wt
        return e[1]

    def get_trainable_padded(self, w, n_pad):
        """W^T of a trainable (N,K) weight as the first N columns of a persistent zero-initialised (K, n_pad) matrix (n_pad >= N): the B
        operand of the data gradient of a Linear layer whose OUTPUT was computed n_pad columns wide (the 50-class layer of the segmentation
        head in a 52-column matrix: _LinearPadN).  Refreshed with the other copies by refresh_trainable."""
        key = ("tp", w.data_ptr(), tuple(w.shape), tuple(w.stride()), int(n_pad))
        e = self.trainable.get(key)
        if e is None or e[0]() is None:
            full = torch.zeros((w.shape[1], int(n_pad)), dtype=w.dtype, device=w.device)
            ops.transpose(w.detach(), out=full[:, :w.shape[0]])
            self.trainable[key] = [weakref.ref(w), full[:, :w.shape[0]], False, full]
            return full
        return e[3]

    def refresh_trainable(self):
        pairs, strided = [], []
        for key, e in list(self.trainable.items()):
            w = e[0]()
            if w is None:
                del self.trainable[key]
            elif e[2]:
                pairs.append((w.detach(), e[1]))
            else:
                strided.append((w, e[1]))
        ops.transpose_batched(pairs)
        for w, wt in strided:               # column windows of a parameter (rows not contiguous as a whole): one launch each
            ops.transpose(w.detach(), out=wt)


TRANSPOSED = _TransposedWeights()


def refresh_caches(model=None):
    """Bring every derived copy of FROZEN weights up to date IN PLACE after weights were loaded into a model whose training step is
    already captured in a HIP graph (load_state_dict changes contents, not addresses; the captured launches read the copies):
    the zero-padded first-conv weight of every patch-embedding Encoder, then the W^T copies of the data-gradient GEMMs (which
    include the transposes of those padded weights -- hence the order), then the bf16 plane images of both.
    Also REQUIRED after any write to a frozen weight that bypasses torch's version counter (`p.data.copy_()`, `p.data.mul_()`, an EMA
    update through `.data`, a custom kernel): the caches key their validity on that counter and cannot see such a write."""
    if model is not None:
        for m in model.modules():
            if hasattr(m, 'refresh_padded_weight'):
                m.refresh_padded_weight()
    TRANSPOSED.refresh()
    ops.PLANES.refresh()              # (after the transposes: the plane images of W^T copies are split from the refreshed copies)


def _wt(w):
    """W^T (K,N) contiguous: cached for frozen weights, a fresh upp_transpose_f32 copy for trainable ones (they change every step)."""
    if w.requires_grad:
        if not (w.is_cuda and w.stride(1) == 1):
            return w.detach().t().contiguous()
        return TRANSPOSED.get_trainable(w) if TRANSPOSED.managed else ops.transpose(w.detach())
    return TRANSPOSED.get(w)


def linear_usable(x, weight):
    """upp_linear_f32 serves f32 HIP operands with 16-byte aligned rows (contraction length a multiple of 4)."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2 and weight.stride(1) == 1
            and weight.stride(0) % 4 == 0 and weight.data_ptr() % 16 == 0
            and x.shape[-1] == weight.shape[1] and weight.shape[1] % 4 == 0 and x.numel() > 0)


def _lin(a, w, bias=None, epi=None, aux=None, dgrad=False):
    """a . op(w)^T with op(w) = w (N,K) -- or, dgrad, w^T: the data gradient dX = dY . W -- on the kernel that serves the weight:
    frozen -> upp_linear_sb_f32 on the cached plane image of w / of its cached transpose (exact-f32 kernel where the split kernel does not
    take the shape); trainable inside a step driver (ops.PLANES.managed) -> upp_linear_sb_f32 on the persistent image that the driver
    re-splits once per step, the transpose read straight from w; trainable elsewhere -> the exact-f32 kernel (a fresh W^T for dgrad)."""
    epi = ops.LIN_NONE if epi is None else epi
    if not w.requires_grad:
        return ops.linear_f32(a, _wt(w) if dgrad else w, bias, epi, aux=aux, frozen=True)
    N, K = (w.shape[1], w.shape[0]) if dgrad else w.shape
    M = a.numel() // a.shape[-1]
    # (the plane image is a buffer of its own, written by upp_linear_sb_prep with scalar reads of w: a column window of a wider weight whose
    #  rows start off a 16-byte boundary -- w[:, 3:] of a (1536, 1155) weight -- is served where it lies)
    if (ops.PLANES.managed and w.is_cuda and w.dim() == 2 and w.stride(1) == 1
            and ops.linear_sb_usable(M, N, K) and (bias is None or bias.data_ptr() % 16 == 0)):
        return ops.linear_f32(a, None, bias, epi, aux=aux, planes=ops.PLANES.get_trainable(w, transposed=dgrad), wshape=(N, K))
    return ops.linear_f32(a, _wt(w) if dgrad else w, bias, epi, aux=aux)


class _LinearMFMA(Function):
    """x . W^T (+ b) on upp_linear_f32, forward and data gradient.  The weight gradient of a TRAINABLE layer is a library GEMM
    (not on the PEFT hot path: the Transformer weights are frozen there); a trainable bias takes the deferred column sum."""

    @staticmethod
    def forward(ctx, x, w, b, own_wgrad=False):
        ctx.save_for_backward(x if w.requires_grad else None, w)
        ctx.bias_ptr = b.data_ptr() if b is not None else 0
        ctx.own_wgrad = own_wgrad
        return _lin(x, w, b, ops.LIN_BIAS if b is not None else ops.LIN_NONE)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if w.shape[0] % 4 == 0 and g2.stride(0) % 4 == 0:
                gx = _lin(g2, w, dgrad=True).view(g.shape[:-1] + (w.shape[1],))
            elif w.shape[0] <= 64 and w.shape[1] <= 256:
                # a narrow layer (the 64 -> 3 score head of the denoising prompter): the data gradient contracts over its few outputs
                gx = ops.linear_smallk(g2, _wt(w), None, 0).view(g.shape[:-1] + (w.shape[1],))
            else:
                note_declined("linear data gradient", "N = %d is not a multiple of 4" % w.shape[0])
                gx = torch.mm(g2, w).view(g.shape[:-1] + (w.shape[1],))
        if ctx.needs_input_grad[1]:
            gw = weight_grad(g2, x.reshape(-1, x.shape[-1]), w, ctx.own_wgrad)
        if b_needed(ctx):
            _, gb = _DEFERRED.reduce(ctx.bias_ptr, g2, 0, g2.shape[1])
        return gx, gw, gb, None


def weight_grad(g2, x2, w, own=False):
    """dW = g2^T . x2 for the trainable weight w (N,K) on upp_linear_wgrad_grouped_f32: inside a training step the pair is queued
    and every weight gradient of the backward pass runs in one launch at the end, its partials summed in split order straight
    into w's slot of the flat gradient buffer (then None is returned); otherwise a group of one, summed at once."""
    N, K = g2.shape[1], x2.shape[1]
    if N % 4 or K % 4 or not g2.is_cuda:
        if N * K <= 2048 and N + K <= 512 and x2.stride(1) == 1 and g2.stride(1) == 1 and g2.is_cuda:
            part = ops.linear_smallk_wgrad(g2, x2)
            _, gw = _DEFERRED.reduce(w.data_ptr(), part.view(part.shape[0], N * K), 0, N * K)
            return None if gw is None else gw.view(N, K)
        if g2.is_cuda:
            note_declined("linear weight gradient (%d,%d)" % (N, K), "N % 4 / K % 4")
        return torch.mm(g2.t(), x2)
    if g2.stride(1) != 1 or g2.stride(0) % 4 or g2.data_ptr() % 16:          # (a misaligned view: one copy makes it servable)
        g2 = g2.contiguous()
    if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    if _DEFERRED.wgrad(w.data_ptr(), g2, x2) or _DEFERRED.wgrad_window(w, g2, x2):
        return None
    part = ops.linear_wgrad(g2, x2)
    return ops.sum_rows(part.view(part.shape[0], -1)).view(part.shape[1:]) if part.shape[0] > 1 else part[0]


def b_needed(ctx):
    return len(ctx.needs_input_grad) > 2 and ctx.needs_input_grad[2]


class _LinearSmallK(Function):
    """x . W^T + b for the layers upp_linear_f32 does not take (K not a multiple of 4 or unaligned rows; K <= 64, N <= 256) WITH a
    gradient: forward upp_linear_smallk_f32, data gradient the same kernel on W^T (contraction over N <= 64), weight gradient
    upp_linear_smallk_wgrad_f32 partials, bias gradient a (two-stage) column sum -- all through the deferred sums of a step driver.
    (The rectify prompter's point-wise layers in the pre-task recipe and stage 2: K = 3, 27, 59.)"""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x if w.requires_grad else None, w)
        ctx.ptrs = (w.data_ptr(), b.data_ptr() if b is not None else 0)
        return ops.linear_smallk(x, w, b, 0)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        N, K = w.shape
        g2 = g.reshape(-1, N)
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if N > 64:      # (_smallk_with_grad: N % 4 == 0, frozen weight) contraction too long for the small kernel: the matrix cores,
                gx = ops.linear_f32(g2, TRANSPOSED.get_padded(w))[:, :K]            # with W^T zero-padded to 4 output columns
            else:
                gx = ops.linear_smallk(g2, w.detach().t().contiguous() if not w.requires_grad else ops.transpose(w.detach()), None, 0)
            gx = gx.reshape(g.shape[:-1] + (K,))
        if ctx.needs_input_grad[1]:
            x2 = x.reshape(-1, K)
            if x2.stride(1) != 1:
                x2 = x2.contiguous()
            part = ops.linear_smallk_wgrad(g2, x2)
            _, gw = _DEFERRED.reduce(ctx.ptrs[0], part.view(part.shape[0], N * K), 0, N * K)
            gw = None if gw is None else gw.view(N, K)
        if ctx.needs_input_grad[2]:
            _, gb = _DEFERRED.reduce(ctx.ptrs[1], g2, 0, N)
        return gx, gw, gb


class _LinearGroupBias(Function):
    """x (M,K) . W^T + gb[m // rows] with the per-group term added in the GEMM's epilogue (upp_linear_group_bias_f32).  Backward: the data
    gradient and the (deferred, grouped) weight gradient of _LinearMFMA; the group term's gradient is the sum of g over each group's rows."""

    @staticmethod
    def forward(ctx, x, w, gb, rows):
        ctx.save_for_backward(x if w.requires_grad else None, w)
        ctx.rows = rows
        if w.requires_grad and ops.PLANES.managed and ops.linear_sb_usable(x.shape[0], w.shape[0], w.shape[1]) and w.data_ptr() % 16 == 0 and w.stride(0) % 4 == 0:
            return ops.linear_group_bias(x, w, gb, rows, planes=ops.PLANES.get_trainable(w))
        return ops.linear_group_bias(x, w, gb, rows, frozen=not w.requires_grad)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g2 = g if g.is_contiguous() else g.contiguous()
        gx = gw = ggb = None
        if ctx.needs_input_grad[0]:
            gx = _lin(g2, w, dgrad=True)
        if ctx.needs_input_grad[1]:
            gw = weight_grad(g2, x, w, True)
        if ctx.needs_input_grad[2]:
            groups = g2.shape[0] // ctx.rows
            if g2.is_cuda and g2.dtype == torch.float32 and groups * ctx.rows == g2.shape[0] and 1 <= groups <= 65535:
                ggb = ops.colsum_partials(g2, 0, g2.shape[1], chunks=groups)        # chunk = group: (groups, N) in one launch, wave-ordered sums
            else:
                ggb = g2.view(groups, ctx.rows, g2.shape[1]).sum(dim=1)
        return gx, gw, ggb, None


class _GroupBiasAdd(Function):
    """y (M,N) + gb[m // rows]: the broadcast add of linear_group_bias when the GEMM cannot take the term in its epilogue (few rows).
    As its own node so that the group term's gradient is upp_colsum_partials (chunk = group) and not torch's sum over the rows of a
    group -- a reduction that splits 2,048 rows over workgroups and zeroes its semaphores with a memset: inside a captured step that
    node is right in the first replay only (NOTEBOOK 12.11; the B = 4 segmentation step of tests/test_gpu_determinism.py)."""

    @staticmethod
    def forward(ctx, y, gb, rows):
        ctx.rows = rows
        M, N = y.shape
        return (y.view(M // rows, rows, N) + gb.unsqueeze(1)).view(M, N)

    @staticmethod
    def backward(ctx, g):
        g2 = g if g.is_contiguous() else g.contiguous()
        ggb = ops.colsum_partials(g2, 0, g2.shape[1], chunks=g2.shape[0] // ctx.rows) if ctx.needs_input_grad[1] else None
        return (g2 if ctx.needs_input_grad[0] else None), ggb, None


class _ExpandRows(Function):
    """token (1,1,D) or (D,) -> (B, N, D) (the mask token of the MAE decoders: `mask_token.expand(B, N, -1)`), with the backward's sum
    over the B N rows on this library's kernels (and routed into the step driver's deferred sums): torch's sum over 1,024 ... 1,216 rows
    per column splits them over workgroups behind a memset -- see _GroupBiasAdd."""

    @staticmethod
    def forward(ctx, token, B, N):
        ctx.ptr, ctx.shape = token.data_ptr(), tuple(token.shape)
        return token.reshape(1, 1, -1).expand(B, N, -1)

    @staticmethod
    def backward(ctx, g):
        D = g.shape[-1]
        g2 = g.reshape(-1, D)
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        if not (g2.is_cuda and g2.dtype == torch.float32):
            return g2.sum(dim=0).view(ctx.shape), None, None
        routed, gt = _DEFERRED.reduce(ctx.ptr, g2, 0, D)
        return (None if routed else gt.view(ctx.shape)), None, None


def expand_rows(token, B, N):
    """token.expand(B, N, -1) whose gradient is summed by this library's kernels (see _ExpandRows)."""
    if token.is_cuda and torch.is_grad_enabled() and token.requires_grad:
        return _ExpandRows.apply(token, int(B), int(N))
    return token.reshape(1, 1, -1).expand(B, N, -1)


def linear_group_bias(x, weight, group_bias, rows_per_group):
    """F.linear(x, weight) + group_bias.repeat_interleave(rows_per_group, 0) for a (M,K) matrix x whose rows come in groups of
    `rows_per_group` (a power of two >= 32) that share one bias row: in the GEMM's epilogue when the problem is tall enough for the
    register-tiled kernel, as a broadcast add otherwise."""
    M, K = x.shape
    N = weight.shape[0]
    r = int(rows_per_group)
    if (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.stride(1) == 1 and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0 and weight.stride(1) == 1 and weight.stride(0) % 4 == 0 and weight.data_ptr() % 16 == 0
            and (K % 4 == 0 and weight.shape[1] == K) and ops.linear_group_bias_usable(M, N, K, r)
            and (not (torch.is_grad_enabled() and x.requires_grad) or K % 4 == 0)):
        gb = group_bias.contiguous()
        if not torch.is_grad_enabled() or not (x.requires_grad or weight.requires_grad or gb.requires_grad):
            return ops.linear_group_bias(x, weight, gb, r, frozen=not weight.requires_grad)
        return _LinearGroupBias.apply(x, weight, gb, r)
    y = linear(x, weight, own_wgrad=True)
    if y.is_cuda and y.dtype == torch.float32 and M % r == 0 and 1 <= M // r <= 65535 and torch.is_grad_enabled() and group_bias.requires_grad:
        return _GroupBiasAdd.apply(y, group_bias, r)
    return (y.view(M // r, r, N) + group_bias.unsqueeze(1)).view(M, N)


def _smallk_with_grad(x, weight):
    """(the data gradient contracts over N on the same kernel: N <= 64 unless the input needs none -- the 3 -> 128 first layer of a
    trainable position MLP reads the centres)"""
    N, K = weight.shape
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2 and x.shape[-1] == K
            and K <= 64 and (N <= 64 or (N <= 256 and (not x.requires_grad or (N % 4 == 0 and not weight.requires_grad))))
            and N * K <= 2048 and N + K <= 512 and x.numel() > 0
            and weight.stride(1) == 1)


_ACT_EPI = {None: (ops.LIN_BIAS, 0), 'relu': (ops.LIN_BIAS_RELU, 1), 'gelu': (ops.LIN_BIAS_GELU, 2)}


def _act_torch(y, act):
    return y if act is None else (F.relu(y) if act == 'relu' else F.gelu(y))


def linear(x, weight, bias=None, own_wgrad=False, act=None):
    """act(F.linear(x, weight, bias)) on this library's kernels: upp_linear_f32 (FP32 matrix cores) for 16-byte aligned rows and
    K % 4 == 0, upp_linear_smallk_f32 (K <= 64, N <= 256, forward only) for the rest; otherwise the library GEMM (said once under
    UPP_VERBOSE).  act: None / 'relu' / 'gelu' (erf) -- fused into the epilogue when nothing needs a gradient, applied by torch
    otherwise.  own_wgrad: kept for callers of round 2 (every weight gradient is ours now, whatever the row count)."""
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad))
    if (x.is_cuda and weight.dim() == 2 and weight.dtype == torch.float32 and weight.shape[1] % 4 == 0 and weight.shape[1] > 64
            and (weight.stride(1) != 1 or weight.stride(0) % 4 or weight.data_ptr() % 16)):
        # a column window of a wider weight whose rows start off a 16-byte boundary (w[:, 3:] of the (1536, 1155) first layer of the
        # segmentation head's feature propagation): inside a step driver a trainable window of a parameter is multiplied through its
        # persistent bf16 plane image and its gradient lands in the parameter's gradient window (_DeferredSums.wgrad_window) -- no copy;
        # anywhere else one differentiable copy makes it servable
        rows = x.numel() // max(x.shape[-1], 1)
        window_ok = bool(needs_grad and weight.requires_grad and ops.PLANES.managed and weight.stride(1) == 1 and x.dtype == torch.float32
                         and isinstance(weight._base, torch.nn.Parameter) and x.shape[-1] == weight.shape[1] and weight.shape[0] % 4 == 0
                         and ops.linear_sb_usable(rows, weight.shape[0], weight.shape[1])
                         and (not x.requires_grad or ops.linear_sb_usable(rows, weight.shape[1], weight.shape[0])))
        if window_ok:
            return _act_torch(_LinearMFMA.apply(x, weight, bias, bool(own_wgrad)), act)
        weight = weight.contiguous()
    if needs_grad and weight.dim() == 2 and weight.shape[0] % 4 and weight.shape[0] > 16 and linear_usable(x, weight):
        # the data gradient contracts over the N outputs and the weight gradient is (N, K): both want N % 4 == 0 (the 50 part classes of
        # the segmentation head's last layer).  Zero rows up to the next multiple of 4 -- their outputs are sliced off, their gradient
        # rows are dropped by the pad's backward -- keep forward and both gradients on our kernels.
        pad = -weight.shape[0] % 4
        y = linear(x, F.pad(weight, (0, 0, 0, pad)), None if bias is None else F.pad(bias, (0, pad)), own_wgrad, act)
        return y[..., :weight.shape[0]]
    if not linear_usable(x, weight):
        if (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2 and x.shape[-1] == weight.shape[1]
                and weight.shape[1] <= 64 and weight.shape[0] <= 256 and x.numel() > 0 and not needs_grad):
            return ops.linear_smallk(x, weight, bias, _ACT_EPI[act][1])
        if needs_grad and _smallk_with_grad(x, weight):
            return _act_torch(_LinearSmallK.apply(x, weight, bias), act)
        if x.is_cuda:
            note_declined("linear %s x %s" % (tuple(x.shape), tuple(weight.shape)), "dtype / layout / K % 4" + (" with a gradient" if needs_grad else ""))
        return _act_torch(F.linear(x, weight, bias), act)
    if not needs_grad:
        if act is not None and bias is None:
            bias = _zero_bias(weight)
        return ops.linear_f32(x, weight, bias, _ACT_EPI[act][0] if bias is not None else ops.LIN_NONE, frozen=not weight.requires_grad)
    return _act_torch(_LinearMFMA.apply(x, weight, bias, bool(own_wgrad)), act)


_ZERO_BIAS = {}


def _zero_bias(weight):
    key = (weight.device, weight.shape[0])
    z = _ZERO_BIAS.get(key)
    if z is None:
        z = _ZERO_BIAS[key] = torch.zeros(weight.shape[0], dtype=torch.float32, device=weight.device)
    return z


class _MlpGelu(Function):
    """fc2( GELU( fc1(x) + b1 ) ) (+ b2) (reference models/Point_MAE_pretask_dev.py:163-168):
    two upp_linear_f32 launches forward (the first carries bias + GELU and stores GELU'), two backward for the data gradient (the
    data gradient of fc2 is multiplied by the saved GELU' in its epilogue, i.e. it IS the gradient at the fc1 pre-activation).
    Frozen weights (the PEFT recipe): the hidden activation is not kept.  Trainable weights (Point-MAE pre-training): it is, and
    dW2 = g^T hid, dW1 = g_z^T x, db1 = column sum of g_z, db2 = column sum of g follow weight_grad / the deferred sums."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2=None):
        hid, d = _lin(x, w1, b1, ops.LIN_BIAS_GELU_D)
        train = w1.requires_grad or w2.requires_grad or b1.requires_grad
        ctx.save_for_backward(d, w1, w2, x if train else None, hid if w2.requires_grad else None)
        ctx.bias_ptrs = (b1.data_ptr(), b2.data_ptr() if b2 is not None else 0)
        return _lin(hid, w2, b2, ops.LIN_BIAS if b2 is not None else ops.LIN_NONE)

    @staticmethod
    def backward(ctx, g):
        d, w1, w2, x, hid = ctx.saved_tensors
        need = ctx.needs_input_grad
        g2 = g.reshape(-1, g.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        g_z = _lin(g2, w2, None, ops.LIN_MUL, aux=d.view(-1, d.shape[-1]), dgrad=True)
        gx = None
        if need[0]:
            gx = _lin(g_z, w1, dgrad=True).view(g.shape[:-1] + (w1.shape[1],))
        gw1 = gb1 = gw2 = gb2 = None
        if need[1]:
            gw1 = weight_grad(g_z, x.reshape(-1, x.shape[-1]), w1)
        if need[2]:
            _, gb1 = _DEFERRED.reduce(ctx.bias_ptrs[0], g_z, 0, g_z.shape[1])
        if need[3]:
            gw2 = weight_grad(g2, hid.reshape(-1, hid.shape[-1]), w2)
        if len(need) > 4 and need[4]:
            _, gb2 = _DEFERRED.reduce(ctx.bias_ptrs[1], g2, 0, g2.shape[1])
        return gx, gw1, gb1, gw2, gb2


# (round 5's residual + LayerNorm fold -- residual add + row-block statistics in the projection GEMM's epilogue, LayerNorm in fc1's A-prologue --
# was measured slower than the row kernel it removed (profiles/r05_fold_ab.txt: 4.63-4.66 against 4.55-4.59 ms per step) and left the library
# with ABI 5; NOTEBOOK section 11.3 keeps the account.)


class _MlpSmallKGelu(Function):
    """lin2(GELU(lin1(x))) for a first layer the small-K kernel serves (K <= 64: the 3 -> 128 layer of a position MLP, reference
    models/Point_MAE_pretask_dev.py:395-399) WITH gradients: GELU and GELU' come out of the first launch (upp_linear_smallk_gelu_d_f32), the
    data gradient of the second layer is multiplied by the saved GELU' in its epilogue -- no torch gelu / gelu_backward pair.  Weight and
    bias gradients as _LinearSmallK / _LinearMFMA (deferred sums inside a step driver)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        hid, d = ops.linear_smallk_gelu_d(x, w1, b1)
        ctx.save_for_backward(x if (w1.requires_grad or x.requires_grad) else None, w1, w2, d, hid if w2.requires_grad else None)
        ctx.ptrs = (w1.data_ptr(), b1.data_ptr() if b1 is not None else 0, b2.data_ptr() if b2 is not None else 0)
        return _lin(hid, w2, b2, ops.LIN_BIAS if b2 is not None else ops.LIN_NONE)

    @staticmethod
    def backward(ctx, g):
        x, w1, w2, d, hid = ctx.saved_tensors
        need = ctx.needs_input_grad
        H, K = w1.shape
        g2 = g.reshape(-1, g.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        g_z = _lin(g2, w2, None, ops.LIN_MUL, aux=d.view(-1, H), dgrad=True)                 # (rows, H) = gradient at the pre-activation
        gx = gw1 = gb1 = gw2 = gb2 = None
        if need[0]:
            if H > 64:
                gx = ops.linear_f32(g_z, TRANSPOSED.get_padded(w1))[:, :K]                   # (_smallk_gelu_usable: w1 frozen then)
            else:
                gx = ops.linear_smallk(g_z, w1.detach().t().contiguous() if not w1.requires_grad else ops.transpose(w1.detach()), None, 0)
            gx = gx.reshape(g.shape[:-1] + (K,))
        if need[1]:
            x2 = x.reshape(-1, K)
            if x2.stride(1) != 1:
                x2 = x2.contiguous()
            part = ops.linear_smallk_wgrad(g_z, x2)
            _, gw1 = _DEFERRED.reduce(ctx.ptrs[0], part.view(part.shape[0], H * K), 0, H * K)
            gw1 = None if gw1 is None else gw1.view(H, K)
        if need[2]:
            _, gb1 = _DEFERRED.reduce(ctx.ptrs[1], g_z, 0, H)
        if need[3]:
            gw2 = weight_grad(g2, hid.reshape(-1, H), w2)
        if need[4]:
            _, gb2 = _DEFERRED.reduce(ctx.ptrs[2], g2, 0, g2.shape[1])
        return gx, gw1, gb1, gw2, gb2


def mlp_smallk_gelu_usable(x, w1, b1, w2, b2):
    """Can _MlpSmallKGelu serve lin2(GELU(lin1(x)))?  The first layer is the small-K kernel's (with the gradient rules of _smallk_with_grad:
    a trainable first weight of more than 64 outputs only when x needs no gradient), its bias exists, the second layer is the matrix-core
    kernels' and the hidden width allows a small-K weight gradient."""
    H = w1.shape[0]
    return (_smallk_with_grad(x, w1) and b1 is not None and H % 4 == 0 and H * w1.shape[1] <= 2048 and w2.dim() == 2 and w2.shape[1] == H
            and w2.dtype == torch.float32 and w2.stride(1) == 1 and w2.stride(0) % 4 == 0 and w2.data_ptr() % 16 == 0 and w2.shape[0] % 4 == 0
            and (b2 is None or b2.data_ptr() % 16 == 0))


def mlp_smallk_gelu(x, w1, b1, w2, b2):
    return _MlpSmallKGelu.apply(x, w1, b1, w2, b2)


def mlp_gelu(x, w1, b1, w2, b2=None):
    """fc2(GELU(fc1(x) + b1)) (+ b2); both GEMMs and the activation on the Linear kernels (split-bf16 for frozen / managed weights)."""
    if not torch.is_grad_enabled() or not (x.requires_grad or w1.requires_grad or w2.requires_grad or b1.requires_grad
                                           or (b2 is not None and b2.requires_grad)):
        hid = ops.linear_f32(x, w1, b1, ops.LIN_BIAS_GELU, frozen=not w1.requires_grad)
        return ops.linear_f32(hid, w2, b2, ops.LIN_BIAS if b2 is not None else ops.LIN_NONE, frozen=not w2.requires_grad)
    return _MlpGelu.apply(x, w1, b1, w2, b2)


# ------------------------------------------------------------------ prompt propagation
class _PropPool(Function):
    """X (B,L',D), absolute neighbour rows i1 (B*G2*8) -> max_k + mean_k of the (drop-path scaled) rows, (B*G2, D)."""

    @staticmethod
    def forward(ctx, X, i1, u, keep):
        X = X.contiguous()
        groups = i1.numel() // 8
        pooled, amax = ops.prop_pool_fwd(X, i1, u, keep, groups)
        ctx.save_for_backward(amax, i1, u)
        ctx.meta = (keep, X.shape)
        return pooled

    @staticmethod
    def backward(ctx, g):
        amax, i1, u = ctx.saved_tensors
        keep, shape = ctx.meta
        return ops.prop_pool_bwd(g.contiguous(), amax, i1, u, keep, shape[0] * shape[1]).view(shape), None, None, None


class _PropInterp(Function):
    """X (B,L',D), lc (B,G2,D) -> X with its last T rows += 0.3 * sum_k w8 * (lc + 0.3 * X[i2])[idx8].  w8 may carry a gradient."""

    @staticmethod
    def forward(ctx, X, lc, i2, idx8, w8):
        X, lc = X.contiguous(), lc.contiguous()
        B, Lp, D = X.shape
        T, G2 = idx8.shape[1], lc.shape[1]
        w8 = w8.contiguous()
        ctx.save_for_backward(i2, idx8, w8, *((X, lc) if ctx.needs_input_grad[4] else ()))
        ctx.meta = (B, Lp, T, G2)
        return ops.prop_interp_fwd(X, lc, i2, idx8, w8, B, Lp, T, G2)

    @staticmethod
    def backward(ctx, g):
        i2, idx8, w8 = ctx.saved_tensors[:3]
        B, Lp, T, G2 = ctx.meta
        g = g.contiguous()
        g_c2, g_X = ops.prop_interp_bwd(g, i2, idx8, w8, B, Lp, T, G2)
        g_w8 = None
        if ctx.needs_input_grad[4]:
            X, lc = ctx.saved_tensors[3:]
            g_w8 = ops.prop_w8_grad(g, X, lc, None, i2, idx8, B, Lp, T, G2)
        return g_X, g_c2.view(B, G2, -1), None, None, g_w8


def prop_pool(X, i1, u=None, keep=1.0):
    return _PropPool.apply(X, i1, u, float(keep))


def prop_interp(X, lc, i2, idx8, w8):
    return _PropInterp.apply(X, lc, i2, idx8, w8)


class PropIndex:
    """Index lists of the propagation step for one forward (they depend only on the centres): absolute neighbour rows
    i1 (groups*8), level-2 centre rows i2 (groups), interpolation neighbours idx8 / weights w8 (B,T,8), and their inverses
    (CSR) for the backward kernels.  Built once per forward and shared by every block."""

    def __init__(self, i1, i2, idx8, w8, rows, csr=None):
        B, T = idx8.shape[0], idx8.shape[1]
        groups = i2.numel()
        self.i1, self.i2, self.idx8, self.w8 = i1, i2, idx8, w8
        self.B, self.T, self.G2, self.rows = B, T, groups // B, rows
        if csr is None:
            self.csr1 = ops.csr_build(i1, rows)
            self.csr2 = ops.csr_build(i2, rows)
            self.csr8 = ops.csr_build(idx8.view(-1), groups, seg_len=T * 8, seg_rows=groups // B)
        else:
            self.csr1, self.csr2, self.csr8 = csr

    def tensors(self):
        """The ten tensors of the index (for handing it from one HIP graph to another through static buffers)."""
        return (self.i1, self.i2, self.idx8, self.w8) + self.csr1 + self.csr2 + self.csr8

    @classmethod
    def from_tensors(cls, t, rows):
        return cls(t[0], t[1], t[2], t[3], rows, csr=((t[4], t[5]), (t[6], t[7]), (t[8], t[9])))


class _PropWeights(Function):
    """The interpolation weights of a PropIndex as a differentiable function of the centres (stage 2 of the recipe: the point cloud, and so
    the centres, carry a gradient back to the prompters).  Forward: the weights the index kernel computed (the reference's sort + 1/(d+eps)
    normalisation, models/Point_MAE_unify.py:33-44); backward: upp_prop_weights_bwd.  The neighbour choice idx8 is piecewise constant."""

    @staticmethod
    def forward(ctx, c1, c2, idx8, w8, eps):
        ctx.save_for_backward(c1, c2, idx8)
        ctx.eps = eps
        return w8.clone()

    @staticmethod
    def backward(ctx, g):
        c1, c2, idx8 = ctx.saved_tensors
        g1, g2 = ops.prop_weights_bwd(c1.contiguous(), c2.contiguous(), idx8, g.contiguous(), ctx.eps)
        return g1, g2, None, None, None


def prop_weights(c1, c2, index, eps=1e-3):
    """index.w8 with the autograd edge to the centres c1 (B,T,3) / c2 (B,G2,3) it was built from."""
    return _PropWeights.apply(c1, c2, index.idx8, index.w8, float(eps))


class _Propagate(Function):
    """Fused pool -> BatchNorm1d -> interpolate of one block (upp_prop_fwd / upp_prop_bwd; upp_prop_w8_grad when w8 carries a gradient)."""

    @staticmethod
    def forward(ctx, X, gamma, beta, index, u, keep, running_mean, running_var, momentum, eps, training, w8):
        X = X.contiguous()
        B, Lp, D = X.shape
        w8 = w8.contiguous()
        out, pooled, amax, mean, rstd = ops.prop_fwd(X, index.i1, u, keep, index.i2, index.idx8, w8, gamma, beta, running_mean,
                                                     running_var, momentum, eps, training, B, Lp, index.T, index.G2)
        ctx.save_for_backward(pooled, amax, mean, rstd, gamma, u, w8, *((X, beta) if ctx.needs_input_grad[11] else ()))
        ctx.meta = (index, keep, training, B, Lp)
        return out

    @staticmethod
    def backward(ctx, g):
        pooled, amax, mean, rstd, gamma, u, w8 = ctx.saved_tensors[:7]
        index, keep, training, B, Lp = ctx.meta
        g = g.contiguous()
        g_X, g_gamma, g_beta = ops.prop_bwd(g, pooled, amax, mean, rstd, gamma, u, keep, w8, index.csr1, index.csr2,
                                            index.csr8, training, B, Lp, index.T, index.G2)
        g_w8 = None
        if ctx.needs_input_grad[11]:
            X, beta = ctx.saved_tensors[7:]
            g_w8 = ops.prop_w8_grad(g, X, pooled, (mean, rstd, gamma, beta), index.i2, index.idx8, B, Lp, index.T, index.G2)
        return (g_X, g_gamma, g_beta) + (None,) * 8 + (g_w8,)


def propagate(X, bn, index, u=None, keep=1.0, training=True, w8=None):
    """X (B,L',D) -> X with its last T rows += 0.3 * interp(bn(pool(X)) + 0.3 * X[i2]); bn: nn.BatchNorm1d (affine).
    w8: the interpolation weights when they carry a gradient (prop_weights), default index.w8 (constants of the forward)."""
    use_batch = training or bn.running_mean is None
    momentum = 0.0 if bn.momentum is None else bn.momentum
    return _Propagate.apply(X, bn.weight, bn.bias, index, u, float(keep), bn.running_mean, bn.running_var, momentum, bn.eps, use_batch,
                            index.w8 if w8 is None else w8)


# ------------------------------------------------------------------ forward-only row operators (frozen prompter branches)
def bn_rows(x, bn, training, relu=False):
    """BatchNorm(+ReLU) over the rows of a channels-last (R,C) matrix; no autograd (frozen branches only)."""
    use_batch = training or bn.running_mean is None
    momentum = 0.0 if bn.momentum is None else bn.momentum
    return ops.bn_rows_fwd(x.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, bn.eps, use_batch, relu)


class _BnRowsTrain(Function):
    """Training-mode BatchNorm(+ReLU)(+Dropout) over rows with its backward (upp_bn_rows_fwd / upp_bn_rows_bwd, upp_bn_rows_drop_*): the
    trainable BatchNorm1d layers of the per-point heads.  The running statistics are updated in place by the forward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, drop_p=0.0, seed=None, salt=0, bump_pending=False):
        x = x.contiguous()
        # the mask is a hash of the counter's value for THIS forward: when the host bumps its counters at the end of the forward
        # (upp_layers.end_forward) the forward reads the old value + 1 and the backward, which runs behind the bump, the new value
        drop = (drop_p, seed, 1 if bump_pending else 0, salt) if drop_p > 0.0 else None
        y, mean, rstd = ops.bn_rows_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, True, relu, want_stats=True, drop=drop)
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.relu, ctx.drop = relu, (None if drop is None else (drop_p, seed, 0, salt))
        return y

    @staticmethod
    def backward(ctx, g):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        g_x, g_gamma, g_beta = ops.bn_rows_bwd(x, g.contiguous(), mean, rstd, gamma, beta, ctx.relu, want_gx=ctx.needs_input_grad[0], drop=ctx.drop)
        return (g_x, g_gamma if ctx.needs_input_grad[1] else None, g_beta if ctx.needs_input_grad[2] else None,
                None, None, None, None, None, None, None, None, None)


def bn_rows_train(x, bn, relu=False, drop_p=0.0, salt=0, bump_pending=False):
    """Differentiable training-mode BatchNorm(+ReLU) over the rows of a channels-last (R,C) matrix; drop_p > 0: nn.Dropout(drop_p) on the
    output in the same passes, its masks a hash of (bn.num_batches_tracked, salt, element) -- the caller bumps the counter once per forward
    (upp_layers.bump_counter, as nn.BatchNorm1d does); bump_pending: that bump has been queued for the end of the forward, not applied."""
    momentum = 0.0 if bn.momentum is None else bn.momentum
    if drop_p > 0.0:
        return _BnRowsTrain.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, bn.eps, relu, float(drop_p),
                                  bn.num_batches_tracked, int(salt), bool(bump_pending))
    return _BnRowsTrain.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, bn.eps, relu)


class _FanOut(Function):
    """n aliases of one tensor whose gradients come back TOGETHER: autograd sums the gradients of a tensor that n consumers read with
    n - 1 element-wise launches (`pos` is added in front of every Transformer block: 11 launches of stage 2's backward); here the n
    gradients are added in branch order by one upp_batched_sum launch (n jobs of one row sharing the destination)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        first = live[0]
        if not (first.is_cuda and first.dtype == torch.float32 and all(g.shape == first.shape and g.dtype == first.dtype for g in live)):
            total = live[0]
            for g in live[1:]:
                total = total + g
            return total, None
        out = torch.empty(first.shape, dtype=torch.float32, device=first.device)
        n = out.numel()
        ops.batched_sum([(g.contiguous().view(1, n), 0, 1, n, n, out, False) for g in live])       # (one group: dst = sum of its jobs, in order)
        return out, None


def fan_out(x, n):
    """-> n tensors equal to x (views of it); use one per consumer.  With a gradient flowing (HIP f32) the n gradients are summed in one
    launch; otherwise x itself is handed out n times."""
    if n > 1 and torch.is_grad_enabled() and x.requires_grad and x.is_cuda and x.dtype == torch.float32 and n <= 64:
        return _FanOut.apply(x, int(n))
    return (x,) * n


class _GroupMax(Function):
    """max over the k rows of every group (upp_group_max_fwd) with the one-pass backward (upp_group_max_bwd)."""

    @staticmethod
    def forward(ctx, x):
        out, amax = ops.group_max_fwd(x)
        ctx.save_for_backward(amax)
        ctx.k = x.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        (amax,) = ctx.saved_tensors
        return ops.group_max_bwd(g.contiguous(), amax, ctx.k)


def group_max(x):
    """x (..., k, C) -> (..., C): max over the second-to-last dimension.  f32 HIP tensors with C % 4 == 0, k <= 255 and contiguous rows take
    the kernel pair (forward: max + arg-max in one read; backward: the gradient tensor written once, no zero-fill + scatter); anything else
    `x.max(dim=-2)[0]`."""
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and x.is_contiguous() and x.shape[-1] % 4 == 0 and 1 <= x.shape[-2] <= 255
            and x.numel() > 0 and x.data_ptr() % 16 == 0):
        lead = tuple(x.shape[:-2])
        x3 = x.reshape(-1, x.shape[-2], x.shape[-1])
        out = _GroupMax.apply(x3) if (torch.is_grad_enabled() and x.requires_grad) else ops.group_max_fwd(x3)[0]
        return out.view(lead + (x.shape[-1],))
    return x.max(dim=-2)[0]


def argsort_rows(key, descending=False, stable=True):
    """torch.argsort(key, dim=-1, descending=..., stable=True) of short rows (N <= 16384) on the rank-counting kernel (upp_argsort_rows)
    for f32 HIP tensors; anything else (CPU tensors on a GPU-less host, other dtypes, long rows) takes torch.argsort.  Integer / bool
    keys of small magnitude (a mask) are ranked through their exact f32 image.  The kernel's order IS the stable one (ties by index, NaN
    above +inf as torch ranks it), so `stable=False` -- which only permits another tie order -- is served by the same launch."""
    if isinstance(key, torch.Tensor) and key.is_cuda and key.dim() >= 1 and 1 <= key.shape[-1] <= 16384 and key.numel() < 2 ** 31:
        if key.dtype == torch.float32:
            return ops.argsort_rows(key.detach(), descending)
        if key.dtype in (torch.bool, torch.uint8, torch.int8, torch.int16):          # exact in f32
            return ops.argsort_rows(key.to(torch.float32), descending)
    return torch.argsort(key, dim=-1, descending=descending, stable=stable)


def sqdist_topk(xyz1, xyz2, k):
    """The k nearest xyz2 points of every xyz1 point (reference square_distance form), ascending; no autograd."""
    return ops.sqdist_topk(xyz1.contiguous(), xyz2.contiguous(), k)


def interp(dists, idx, feat, k, eps, out=None, col0=0):
    """Inverse-distance interpolation from the k nearest of a sorted neighbour table; no autograd."""
    return ops.interp_fwd(dists, idx, feat.contiguous(), k, eps, out, col0)


class _InterpTrain(Function):
    """Inverse-distance interpolation with a gradient for the features (upp_interp_fwd / upp_interp_bwd); the sorted
    neighbour table (dists, idx) is a constant of the graph."""

    @staticmethod
    def forward(ctx, dists, idx, feat, k, eps):
        feat = feat.contiguous()
        ctx.save_for_backward(dists, idx)
        ctx.S, ctx.k, ctx.eps = feat.shape[1], k, eps
        return ops.interp_fwd(dists, idx, feat, k, eps)

    @staticmethod
    def backward(ctx, g):
        dists, idx = ctx.saved_tensors
        return None, None, ops.interp_bwd(dists, idx, g.contiguous(), ctx.S, ctx.k, ctx.eps), None, None


_UNIT_DIST = {}


def gather_rows(points, idx):
    """points (B,S,C) f32, idx (B,M) int64 per-sample row indices -> (B,M,C) = points[b, idx[b, m]] with a DETERMINISTIC, sort-free
    backward: a gather is the k = 1 inverse-distance interpolation (its single weight is (1/(d+eps)) / (1/(d+eps)) = 1 exactly), so the
    forward is upp_interp_fwd and the gradient the pull kernel of upp_interp_bwd -- torch's advanced-indexing backward is an
    index_put_(accumulate=True) behind a device radix sort (reference models/Point_MAE_pretask_dev.py:409-413 set abstraction, whose
    features carry a gradient in the pre-task and stage-2 recipes).  M <= 4096; anything else takes torch.gather."""
    B, S, C = points.shape
    M = idx.shape[1]
    if not (points.is_cuda and points.dtype == torch.float32 and idx.dtype == torch.int64 and idx.dim() == 2 and 0 < M <= 4096 and S > 0):
        return torch.gather(points, 1, idx.unsqueeze(-1).expand(-1, -1, C))
    key = (B, M, str(points.device))
    d = _UNIT_DIST.get(key)
    if d is None:
        d = torch.zeros(B, M, 1, device=points.device)
        if not torch.cuda.is_current_stream_capturing():          # (memory of a graph's private pool is never cached)
            # never evicted (B M 4 bytes per shape): a HIP graph captured earlier has the address of a cached tensor baked into its interp
            # launches -- returning that memory to the allocator would let a later tensor put anything under a replayed graph's weights
            _UNIT_DIST[key] = d
    tab = idx.contiguous().view(B, M, 1)
    if torch.is_grad_enabled() and points.requires_grad:
        return _InterpTrain.apply(d, tab, points, 1, 1.0)
    return ops.interp_fwd(d, tab, points.contiguous(), 1, 1.0)


def interp_train(dists, idx, feat, k, eps):
    """Differentiable (w.r.t. feat) inverse-distance interpolation from the k nearest of a sorted neighbour table."""
    return _InterpTrain.apply(dists, idx, feat, k, eps)


class _InterpGeo(Function):
    """Inverse-distance interpolation with gradients for the features AND the geometry (queries and sources): forward upp_sqdist_topk +
    upp_interp_fwd, backward upp_interp_bwd (features) + upp_interp_geo_bwd (coordinates) -- the autograd chain of the reference's
    square_distance / sort / index / weights formulation (models/Point_MAE_unify.py:22-48) in five launches."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, feat, k, eps):
        xyz1, xyz2, feat = xyz1.contiguous(), xyz2.contiguous(), feat.contiguous()
        dists, idx = ops.sqdist_topk(xyz1, xyz2, k)
        ctx.save_for_backward(xyz1, xyz2, feat, dists, idx)
        ctx.k, ctx.eps = k, eps
        return ops.interp_fwd(dists, idx, feat, k, eps)

    @staticmethod
    def backward(ctx, g):
        xyz1, xyz2, feat, dists, idx = ctx.saved_tensors
        g = g.contiguous()
        need = ctx.needs_input_grad
        g_feat = ops.interp_bwd(dists, idx, g, feat.shape[1], ctx.k, ctx.eps) if need[2] else None
        g1 = g2 = None
        if need[0] or need[1]:
            g1, g2 = ops.interp_geo_bwd(dists, idx, feat, g, xyz1, xyz2, ctx.k, ctx.eps, need[0], need[1])
        return g1, g2, g_feat, None, None


def interp_geo(xyz1, xyz2, feat, k, eps):
    """(B,N,C) inverse-distance interpolation of feat (B,S,C) at xyz1 from its k nearest of xyz2, differentiable w.r.t. all three."""
    return _InterpGeo.apply(xyz1, xyz2, feat, int(min(k, xyz2.shape[1])), float(eps))


class _InterpAffine(Function):
    """interp(feat) + x3 . wt in one launch (upp_interp_affine_fwd); gradients for feat (upp_interp_bwd) and wt (one skinny
    GEMM); the neighbour table and x3 are constants of the graph."""

    @staticmethod
    def forward(ctx, dists, idx, feat, x3, wt, k, eps):
        feat, x3, wt = feat.contiguous(), x3.contiguous(), wt.contiguous()
        ctx.save_for_backward(dists, idx, x3)
        ctx.S, ctx.k, ctx.eps = feat.shape[1], k, eps
        return ops.interp_affine_fwd(dists, idx, feat, x3, wt, k, eps)

    @staticmethod
    def backward(ctx, g):
        dists, idx, x3 = ctx.saved_tensors
        g = g.contiguous()
        g_feat = ops.interp_bwd(dists, idx, g, ctx.S, ctx.k, ctx.eps) if ctx.needs_input_grad[2] else None
        g_wt = None
        if ctx.needs_input_grad[4]:
            g2, x2 = g.reshape(-1, g.shape[-1]), x3.reshape(-1, 3)
            if g2.shape[1] * 3 <= 2048 and g2.shape[1] + 3 <= 512:           # (3, C) = x3^T g over all points: the small-K weight-gradient kernel (transposed roles)
                g_wt = ops.linear_smallk_wgrad(g2, x2).sum(0).t()
            elif g2.shape[1] % 4 == 0 and g2.stride(0) % 4 == 0 and g2.data_ptr() % 16 == 0 and g2.is_cuda:
                # wide rows (C = 1536 over 65,536 label points): three weighted column sums of g, read once
                g_wt = ops.wcolsum_partials(g2, x2).sum(0)
            else:
                note_declined("rank-3 weight gradient (3,%d)" % g2.shape[1], "row alignment")
                g_wt = torch.mm(x2.t(), g2)
        return None, None, g_feat, None, g_wt, None, None


def interp_affine_train(dists, idx, feat, x3, wt, k, eps):
    """Differentiable (w.r.t. feat and wt) interpolation + rank-3 term: (B,N,C) = interp(feat) + x3 (B,N,3) . wt (3,C)."""
    return _InterpAffine.apply(dists, idx, feat, x3, wt, k, eps)


def posenc(x, freqs, out=None, col0=0):
    """(x, sin(f x), cos(f x))_f positional embedding of (...,3) coordinates; no autograd."""
    return ops.posenc_fwd(x.contiguous(), freqs, out, col0)


# ------------------------------------------------------------------ per-point segmentation tail
class _LinearPadN(Function):
    """y (M, Np) = x . Wp^T for Wp = the (N,K) weight W zero-padded to Np = ceil4(N) rows, WITHOUT materialising Wp (reference
    models/Point_MAE_unify_segment.py:432 the 50-class Conv1d; the GEMM kernels want N % 4 == 0): the forward multiplies through W's own
    bf16 plane image (upp_linear_sb_prep zero-fills beyond N), the data gradient through a persistent zero-padded W^T (K, Np), and the
    weight gradient's partial tiles are (Np, K) of which the first N rows are summed into W's gradient.  F.pad of weight and bias (two
    fills + two copies per step), the [:, :N] slice of the output and its backward (a zero-fill + a copy of the (M, Np) matrix) are gone."""

    @staticmethod
    def forward(ctx, x, w):
        N, K = w.shape
        Np = (N + 3) // 4 * 4
        planes = ops.PLANES.get_trainable(w) if w.requires_grad else ops.PLANES.get(w)
        ctx.save_for_backward(x if w.requires_grad else None, w)
        ctx.Np = Np
        return ops.linear_f32(x, None, None, ops.LIN_NONE, planes=planes, wshape=(Np, K))

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        N, K = w.shape
        g = g if g.is_contiguous() else g.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            wt = (TRANSPOSED.get_trainable_padded(w, ctx.Np) if w.requires_grad and TRANSPOSED.managed
                  else F.pad(w.detach().t(), (0, ctx.Np - N)).contiguous())           # (outside a step driver: a fresh padded copy)
            gx = ops.linear_f32(g, wt)
        if ctx.needs_input_grad[1]:
            if not _DEFERRED.wgrad_rows(w.data_ptr(), g, x, N):
                part = ops.linear_wgrad(g, x)
                gw = (part.sum(dim=0) if part.shape[0] > 1 else part[0])[:N]
        return gx, gw


def linear_pad_n_usable(x, w):
    """Can _LinearPadN serve F.linear(x, w) for an (N,K) weight with N % 4 != 0?  HIP f32 rows, the split-bf16 kernel takes (M, ceil4(N), K),
    and the weight is frozen or managed by a step driver (persistent plane image and padded W^T, refreshed once per step)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and w.dim() == 2 and w.dtype == torch.float32 and w.stride(1) == 1 and x.shape[1] == w.shape[1] and w.shape[0] % 4 != 0):
        return False
    if w.requires_grad and torch.is_grad_enabled() and not (ops.PLANES.managed and TRANSPOSED.managed and isinstance(w._base if w._base is not None else w, torch.nn.Parameter)):
        return False
    if not w.requires_grad and torch.is_grad_enabled() and x.requires_grad:
        return False            # (a frozen weight under a gradient for x: the padded frozen W^T is not kept -- the padded-weight path serves it)
    return ops.linear_sb_usable(x.shape[0], (w.shape[0] + 3) // 4 * 4, w.shape[1])


class _LogSoftmaxRows(Function):
    """log_softmax(y[:, :C] + bias) over the rows of a (R, Cpad) matrix (upp_logsoftmax_rows_fwd / _bwd): the gradient comes back in the
    padded layout, pad columns zero; a trainable bias takes the deferred column sum of that gradient."""

    @staticmethod
    def forward(ctx, y, bias, C):
        logp = ops.logsoftmax_rows_fwd(y, bias, C)
        ctx.save_for_backward(logp)
        ctx.Cpad, ctx.C = y.shape[1], C
        ctx.bias_ptr = bias.data_ptr() if bias is not None else 0
        return logp

    @staticmethod
    def backward(ctx, g):
        (logp,) = ctx.saved_tensors
        g_y = ops.logsoftmax_rows_bwd(g if g.is_contiguous() else g.contiguous(), logp, ctx.Cpad)
        gb = None
        if ctx.bias_ptr and ctx.needs_input_grad[1]:
            _, gb = _DEFERRED.reduce(ctx.bias_ptr, g_y, 0, ctx.C)
        return g_y, gb, None


def log_softmax_rows(y, bias, C):
    """F.log_softmax(y[:, :C] + bias, dim=-1) for a 2-D f32 HIP matrix y (R, >= C), C <= 64."""
    return _LogSoftmaxRows.apply(y, bias, int(C))


class _NllMean(Function):
    """F.nll_loss(logp, target) (mean reduction) in two launches forward, one backward (upp_nll_mean_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, logp, target):
        ctx.save_for_backward(target)
        ctx.shape = tuple(logp.shape)
        return ops.nll_mean_fwd(logp, target).view(())

    @staticmethod
    def backward(ctx, g):
        (target,) = ctx.saved_tensors
        return ops.nll_mean_bwd(g.reshape(1).contiguous(), target, ctx.shape[0], ctx.shape[1]), None


def nll_mean(logp, target):
    """F.nll_loss(logp, target) for contiguous f32 HIP log-probabilities (R, C) and int64 targets (R); torch elsewhere."""
    if (logp.is_cuda and logp.dtype == torch.float32 and logp.dim() == 2 and logp.is_contiguous() and target.dtype == torch.int64
            and target.is_cuda and target.numel() == logp.shape[0]):
        return _NllMean.apply(logp, target.contiguous().view(-1))
    return F.nll_loss(logp, target)


class _NoiseLoss(Function):
    """(positive + negative, score) of the pre-task noise supervision in two launches forward, one backward (upp_noise_loss_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, pred, noise_vector, pn):
        pred, noise_vector = pred.contiguous(), noise_vector.contiguous()
        loss, score = ops.noise_loss_fwd(pred, noise_vector, pn)
        ctx.save_for_backward(pred, noise_vector)
        ctx.pn = pn
        ctx.mark_non_differentiable(score)
        ctx.set_materialize_grads(False)
        return loss.view(()), score

    @staticmethod
    def backward(ctx, g, _g_score):
        if g is None:
            return None, None, None
        pred, nv = ctx.saved_tensors
        return ops.noise_loss_bwd(g.reshape(1).contiguous(), pred, nv, ctx.pn), None, None


def noise_loss(pred, noise_vector, point_num):
    """mean |pred_noise - noise_vector|^2 + mean |pred_pure|^2 and score = |pred| (reference models/Point_MAE_pretask_dev.py:685-692, a
    3-channel rectify prompter): pred (B,P,3) = [point_num shape points | noise points].  -> (loss, score (B,P))."""
    return _NoiseLoss.apply(pred, noise_vector, int(point_num))


# ------------------------------------------------------------------ classification tail
class _ClsPool(Function):
    """feat = [LN(x)[:, 0] | max over rows 1.. of LN(x)]  (upp_cls_pool_fwd / bwd); LayerNorm parameters get no gradient."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        feat, amax, mean, rstd = ops.cls_pool_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, mean, rstd, gamma, amax)
        return feat

    @staticmethod
    def backward(ctx, g_feat):
        x, mean, rstd, gamma, amax = ctx.saved_tensors
        return ops.cls_pool_bwd(g_feat.contiguous(), x, mean, rstd, gamma, amax), None, None, None


def cls_pool(x, norm):
    """torch.cat([norm(x)[:, 0], norm(x)[:, 1:].max(1)[0]], -1) for a frozen nn.LayerNorm `norm`."""
    return _ClsPool.apply(x, norm.weight, norm.bias, float(norm.eps))


class _CrossEntropyAcc(Function):
    """(mean cross-entropy, top-1 accuracy * 100) of logits (B,C) against int64 labels in one launch (upp_ce_acc)."""

    @staticmethod
    def forward(ctx, logits, labels):
        out2, dlogits = ops.ce_acc(logits.contiguous(), labels.contiguous())
        ctx.save_for_backward(dlogits)
        loss, acc = out2[0], out2[1]
        ctx.mark_non_differentiable(acc)
        ctx.set_materialize_grads(False)          # (no zero scalar for the accuracy's "gradient")
        return loss, acc

    @staticmethod
    def backward(ctx, g_loss, _g_acc):
        if g_loss is None:
            return None, None
        (dlogits,) = ctx.saved_tensors
        return dlogits * g_loss, None


class _BnReluDrop(Function):
    """Dropout(ReLU(BatchNorm1d(z))) of a few-row matrix in one launch each way (upp_bn_relu_drop_fwd / bwd)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, momentum, eps, training, u, p):
        z = z.contiguous()
        a, mean, rstd = ops.bn_relu_drop_fwd(z, gamma, beta, running_mean, running_var, momentum, eps, training, u, p)
        ctx.save_for_backward(z, gamma, beta, mean, rstd, u)
        ctx.meta = (training, p)
        return a

    @staticmethod
    def backward(ctx, g_a):
        z, gamma, beta, mean, rstd, u = ctx.saved_tensors
        training, p = ctx.meta
        g_z, g_gamma, g_beta = ops.bn_relu_drop_bwd(g_a.contiguous(), z, gamma, beta, mean, rstd, training, u, p)
        return (g_z, g_gamma, g_beta) + (None,) * 7


def bn_relu_drop(z, bn, u=None, p=0.0, training=True):
    """dropout_p(relu(bn(z))) for a (rows, C) matrix with few rows; u: uniforms (rows, C) when p > 0 in training."""
    use_batch = training or bn.running_mean is None
    momentum = 0.0 if bn.momentum is None else bn.momentum
    return _BnReluDrop.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, momentum, bn.eps, use_batch, u, float(p))


def cross_entropy_acc(logits, labels):
    return _CrossEntropyAcc.apply(logits, labels)


class _BiasGelu(Function):
    """GELU(z + bias) for a frozen bias (upp_bias_gelu_fwd / bwd)."""

    @staticmethod
    def forward(ctx, z, bias):
        # (autograd only calls this forward when a gradient is wanted: the derivative is produced in the same pass and z is
        # not kept -- the backward is one multiply per element instead of a second erf + exp)
        h, d = ops.bias_gelu_fwd_d(z.contiguous(), bias)
        ctx.save_for_backward(d)
        return h

    @staticmethod
    def backward(ctx, g_h):
        d, = ctx.saved_tensors
        return g_h * d, None


def bias_gelu(z, bias):
    if not (torch.is_grad_enabled() and z.requires_grad):
        return ops.bias_gelu_fwd(z.contiguous(), bias)          # frozen front-end / evaluation: value only
    return _BiasGelu.apply(z, bias)


# ------------------------------------------------------------------ bottleneck adapter
class _Adapter(Function):
    """out = x + scale * (W2 . dropout(gelu(W1 . ha + b1)) + b2); see upp_adapter_fwd."""

    @staticmethod
    def forward(ctx, ha, x, W1, b1, W2, b2, u, p, scale):
        ha, x = ha.contiguous(), x.contiguous()
        out, s1 = ops.adapter_fwd(ha, x, W1, b1, W2, b2, u, p, scale)
        ctx.save_for_backward(ha, s1, W1, W2, u)
        ctx.meta = (p, scale)
        ctx.param_ptrs = (W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr())
        return out

    @staticmethod
    def backward(ctx, g_out):
        ha, s1, W1, W2, u = ctx.saved_tensors
        p, scale = ctx.meta
        g_out = g_out.contiguous()
        g_ha, part = ops.adapter_bwd(g_out, ha, s1, W1, W2, u, p, scale)
        need = ctx.needs_input_grad
        gW1 = gb1 = gW2 = gb2 = None
        if need[2] or need[3] or need[4] or need[5]:
            H, D = W1.shape
            pW1, pb1, pW2, pb2 = ctx.param_ptrs
            # part (workgroups, [dW1 | dW2 | db1 | db2]); rows are summed in workgroup order: deterministic
            _, gW1 = _DEFERRED.reduce(pW1, part, 0, H * D)
            _, gW2 = _DEFERRED.reduce(pW2, part, H * D, H * D)
            _, gb1 = _DEFERRED.reduce(pb1, part, 2 * H * D, H)
            _, gb2 = _DEFERRED.reduce(pb2, part, 2 * H * D + H, D)
            gW1 = gW1.view(H, D) if gW1 is not None else None
            gW2 = gW2.view(D, H) if gW2 is not None else None
        return (g_ha if need[0] else None, g_out if need[1] else None, gW1 if need[2] else None, gb1 if need[3] else None,
                gW2 if need[4] else None, gb2 if need[5] else None, None, None, None)


FUSE_TAIL_BACKWARD = True     # _LnAdapter.backward: one launch (upp_ln_adapter_bwd_fused) instead of adapter backward + row backward
ADAPTER_FACTORS = True        # ... which inside a deferred scope writes per-row factors; the weight gradients of all blocks in one launch at its exit


class _LnAdapter(Function):
    """The tail of a block in one launch: rows = strip(x + dp_scale(u) (y + ybias)), out = rows + scale * adapter(LayerNorm(rows)).
    Forward: upp_ln_adapter_fwd.  Backward: upp_ln_adapter_bwd (the adapter's, rebuilding the LayerNorm output from the saved
    rows) followed by upp_rowln_bwd -- the two kernels of the unfused path, minus the stored LayerNorm output."""

    @staticmethod
    def forward(ctx, x, y, ybias, u, keep, mode, P, gamma, beta, eps, W1, b1, W2, b2, ud, pd, scale):
        x = x.contiguous()
        B, Lin, D = x.shape
        Lout = Lin - P if mode in (ROW_STRIP_CLS, ROW_STRIP) else Lin
        y_c = y.contiguous() if y is not None else None
        out, xo, mean, rstd, s1 = ops.ln_adapter_fwd(x, y_c, ybias, u, keep, mode, P, gamma, beta, eps, W1, b1, W2, b2, ud, pd, scale, Lout)
        ctx.save_for_backward(xo, mean, rstd, gamma, beta, u, s1, W1, W2, ud)
        ctx.dims = (B, Lin, Lout, D, P, mode)
        ctx.meta = (keep, pd, scale, y is not None)
        ctx.param_ptrs = (gamma.data_ptr(), beta.data_ptr(), W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr())
        return out

    @staticmethod
    def backward(ctx, g_out):
        xo, mean, rstd, gamma, beta, u, s1, W1, W2, ud = ctx.saved_tensors
        B, Lin, Lout, D, P, mode = ctx.dims
        keep, pd, scale, has_y = ctx.meta
        need = ctx.needs_input_grad
        g_out = g_out.contiguous()
        need_ln = need[7] or need[8]
        need_ad = need[10] or need[11] or need[12] or need[13]
        p_gamma, p_beta, pW1, pb1, pW2, pb2 = ctx.param_ptrs
        H = W1.shape[0]
        # inside a deferred scope (TrainStep) with all four adapter buffers registered: per-row factors now, the weight gradients of every
        # block in one launch at scope exit
        factors = (FUSE_TAIL_BACKWARD and ADAPTER_FACTORS and need[10] and need[11] and need[12] and need[13]
                   and _DEFERRED.targets is not None and all(_DEFERRED.targets.get(p_) is not None for p_ in (pW1, pb1, pW2, pb2)))
        if factors:
            g_x, g_y, fac, ln_part = ops.ln_adapter_bwd_fused(g_out, xo, mean, rstd, gamma, beta, s1, W1, W2, ud, pd, scale, u, keep, mode, P, Lin,
                                                              need[0], has_y and need[1], True, need_ln, factors=True)
            R = B * Lout
            factors = _DEFERRED.adapter((pW1, pW2, pb1, pb2), (H * D, H * D, H, D),
                                        (xo.view(R, D), mean.view(R), rstd.view(R), gamma, beta, g_out.view(R, D), fac, scale))
            if not factors:
                raise RuntimeError("ln_adapter backward: the registered adapter gradient buffers do not match the adapter's shapes")
            part = None
        elif FUSE_TAIL_BACKWARD:
            g_x, g_y, part, ln_part = ops.ln_adapter_bwd_fused(g_out, xo, mean, rstd, gamma, beta, s1, W1, W2, ud, pd, scale, u, keep, mode, P, Lin,
                                                               need[0], has_y and need[1], need_ad, need_ln)
        else:       # the two kernels of the unfused path (kept: tests compare the fused launch against them)
            g_ha, part = ops.ln_adapter_bwd(g_out, xo, mean, rstd, gamma, beta, s1, W1, W2, ud, pd, scale)
            g_x, _, g_y, ln_part = ops.rowln_bwd(g_out, g_ha, xo, mean, rstd, gamma, mode, u, keep, B, Lin, Lout, D, P,
                                                 need_x=need[0], need_prompt=False, need_y=has_y and need[1], need_ln_part=need_ln)
        g_gamma = g_beta = gW1 = gb1 = gW2 = gb2 = None
        if need_ln:
            _, g_gamma = _DEFERRED.reduce(p_gamma, ln_part, 0, D)
            _, g_beta = _DEFERRED.reduce(p_beta, ln_part, D, D)
        if need_ad and not factors:
            _, gW1 = _DEFERRED.reduce(pW1, part, 0, H * D)
            _, gW2 = _DEFERRED.reduce(pW2, part, H * D, H * D)
            _, gb1 = _DEFERRED.reduce(pb1, part, 2 * H * D, H)
            _, gb2 = _DEFERRED.reduce(pb2, part, 2 * H * D + H, D)
            gW1 = gW1.view(H, D) if gW1 is not None else None
            gW2 = gW2.view(D, H) if gW2 is not None else None
        return (g_x if need[0] else None, g_y, None, None, None, None, None, g_gamma if need[7] else None, g_beta if need[8] else None,
                None, gW1 if need[10] else None, gb1 if need[11] else None, gW2 if need[12] else None, gb2 if need[13] else None,
                None, None, None)


# (round 5 also let this launch compute the NEXT block's head -- upp_ln_adapter_fwd_next: bit-identical, and slower: the tail grew by what the
# head launch cost and the step went 4.57 -> 4.71 ms; it left the library with ABI 5, NOTEBOOK section 11.6.)
def ln_adapter(x, y, ybias, u, keep, mode, P, ln, W1, b1, W2, b2, ud=None, pd=0.0, scale=0.7):
    """One launch for `rowln(x, y=y, ybias=ybias, u=u, keep=keep, mode=mode, P=P, gamma, beta)` + `adapter(ha, rows, ...)`
    (mode: ROW_IDENTITY / ROW_STRIP_CLS / ROW_STRIP).  Limits: D == 384, 32 hidden units."""
    if ybias is not None and (y is None or (torch.is_grad_enabled() and ybias.requires_grad)):
        raise ValueError("ybias is the frozen bias of the Linear that produced y")
    return _LnAdapter.apply(x, y, ybias, u, float(keep), int(mode), int(P), ln.weight, ln.bias, float(ln.eps), W1, b1, W2, b2, ud,
                            float(pd), float(scale))


def adapter(ha, x, W1, b1, W2, b2, u=None, p=0.0, scale=0.7):
    return _Adapter.apply(ha, x, W1, b1, W2, b2, u, float(p), float(scale))
