"""Torch-facing operators over the C ABI (include/upp_hip.h).

Each function validates its tensors the way the reference extensions do
(device, dtype, contiguity, shape), allocates outputs with torch (the library
never allocates), and enqueues the HIP kernels on torch's current stream.
CPU tensors raise: there is no fallback path.
"""
import os
import weakref

import torch

from . import _abi


def _need(t, name, dtype, ndim=None, last=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a HIP (cuda) tensor; upp_hip has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"{name} must have {ndim} dims, got {tuple(t.shape)}")
    if last is not None and t.shape[-1] != last:
        raise RuntimeError(f"{name} last dim must be {last}, got {tuple(t.shape)}")


def _same_device(*ts):
    d = ts[0].device
    for t in ts[1:]:
        if t.device != d:
            raise RuntimeError("all tensors must be on the same device")


def _call(dev, name, *args):
    """Enqueue one C-ABI entry point on `dev`'s current stream; raise on a non-zero status."""
    with torch.cuda.device(dev):
        _abi.check(getattr(_abi.load(), name)(*args, _abi.stream()))


# ------------------------------------------------------------------ FPS / gather
import threading

_FPS_FORM = threading.local()          # .form: bits 8-12 of upp_fps_ex's `waves` (clouds per workgroup, whole-LDS reservation); per THREAD


def _fps_form_bits():
    return getattr(_FPS_FORM, "form", 0)


class fps_form:
    """with ops.fps_form(2, True): ... -- FPS launches of THIS thread inside pack `clouds` (1 / 2 / 4) clouds into a workgroup and, with
    `exclusive`, reserve their CUs' whole LDS.  Same indices.  For a stream that runs BESIDE another one (the pipelined step's front-end): a
    32-cloud FPS that shares 32 CUs with the other stream's workgroups slows every one-round GEMM of that stream for as long as it is resident;
    16 CUs of its own cost the FPS 11 % and return 1.5 % of the step (tools/micro/fps_cpw_ab.sh).  Stand-alone the spread form is the faster
    one: the default.  The form is thread-local: another thread's FPS calls issued meanwhile keep their own."""

    def __init__(self, clouds, exclusive=False):
        if clouds not in (1, 2, 4):
            raise ValueError("clouds per workgroup: 1, 2 or 4")
        self.form = (int(clouds) << 8) | (0x1000 if exclusive else 0)

    def __enter__(self):
        self.prev = _fps_form_bits()
        _FPS_FORM.form = self.form

    def __exit__(self, *exc):
        _FPS_FORM.form = self.prev
        return False


class option:
    """with ops.option("EMBED_SPLIT_BF16", 0): ... -- one of the library's four documented process-wide options (include/upp_hip.h
    upp_set_option: SB_TUNED, SB_XCD2D, STORE_WT, EMBED_SPLIT_BF16) set for the duration of the block; tests and A/B measurements."""

    def __init__(self, name, value):
        self.key, self.value = _abi.OPTIONS[name], int(value)

    def __enter__(self):
        lib = _abi.load()
        self.prev = int(lib.upp_get_option(self.key))
        _abi.check(lib.upp_set_option(self.key, self.value))

    def __exit__(self, *exc):
        _abi.check(_abi.load().upp_set_option(self.key, self.prev))
        return False


def get_option(name):
    return int(_abi.load().upp_get_option(_abi.OPTIONS[name]))


def fps(xyz, npoint, want_centers=False, waves=0):
    """(B,N,3) f32 -> idx (B,npoint) int32 [, centers (B,npoint,3)].  waves: wavefronts per cloud (0 = the library's choice).
    Precondition: finite coordinates (include/upp_hip.h upp_fps: with a NaN in a cloud the indices stay in range but need not be the
    reference kernel's; not checked here -- a check would be a host synchronisation on the hot path)."""
    _need(xyz, "xyz", torch.float32, 3, 3)
    B, N, _ = xyz.shape
    npoint = int(npoint)
    idx = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    centers = torch.empty((B, npoint, 3), dtype=torch.float32, device=xyz.device) if want_centers else None
    if B == 0:                               # an empty batch: nothing to launch (an empty tensor has no device pointer to hand over)
        return (idx, centers) if want_centers else idx
    _call(xyz.device, "upp_fps_ex", _abi.ptr(xyz), _abi.ptr(idx), _abi.ptr(centers), B, N, npoint, int(waves) | _fps_form_bits())
    return (idx, centers) if want_centers else idx


def gather_fwd(features, idx):
    _need(features, "features", torch.float32, 3)
    _need(idx, "idx", torch.int32, 2)
    _same_device(features, idx)
    B, C, N = features.shape
    M = idx.shape[1]
    out = torch.empty((B, C, M), dtype=torch.float32, device=features.device)
    _call(features.device, "upp_gather_fwd", _abi.ptr(features), _abi.ptr(idx), _abi.ptr(out), B, C, N, M)
    return out


def gather_bwd(grad_out, idx, N):
    _need(grad_out, "grad_out", torch.float32, 3)
    _need(idx, "idx", torch.int32, 2)
    B, C, M = grad_out.shape
    grad = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
    _call(grad_out.device, "upp_gather_bwd", _abi.ptr(grad_out), _abi.ptr(idx), _abi.ptr(grad), B, C, N, M)
    return grad


# ------------------------------------------------------------------ kNN / group
def knn(ref, query, k, want_dist=True, want_neigh=False, prefilter=True):
    """ref (B,N,3), query (B,Q,3) -> dist (B,Q,k) f32 | None, idx (B,Q,k) int64 [, neigh (B,Q,k,3)]."""
    _need(ref, "ref", torch.float32, 3, 3)
    _need(query, "query", torch.float32, 3, 3)
    _same_device(ref, query)
    B, N, _ = ref.shape
    Q = query.shape[1]
    if query.shape[0] != B:
        raise RuntimeError("ref and query batch sizes differ")
    k = int(k)
    idx = torch.empty((B, Q, k), dtype=torch.int64, device=ref.device)
    dist = torch.empty((B, Q, k), dtype=torch.float32, device=ref.device) if want_dist else None
    neigh = torch.empty((B, Q, k, 3), dtype=torch.float32, device=ref.device) if want_neigh else None
    if B == 0:
        if k < 1 or k > min(N, 64):
            raise RuntimeError("knn: k must be in [1, min(N, 64)]")
        return dist, idx, neigh
    _call(ref.device, "upp_knn_ex", _abi.ptr(ref), _abi.ptr(query), _abi.ptr(dist), _abi.ptr(idx), _abi.ptr(neigh), B, N, Q, k, int(bool(prefilter)))
    return dist, idx, neigh


def group_fwd(xyz, center, idx):
    _need(xyz, "xyz", torch.float32, 3, 3)
    _need(center, "center", torch.float32, 3, 3)
    _need(idx, "idx", torch.int64, 3)
    _same_device(xyz, center, idx)
    B, N, _ = xyz.shape
    _, G, K = idx.shape
    out = torch.empty((B, G, K, 3), dtype=torch.float32, device=xyz.device)
    _call(xyz.device, "upp_group_fwd", _abi.ptr(xyz), _abi.ptr(center), _abi.ptr(idx), _abi.ptr(out), B, N, G, K)
    return out


def group_bwd(grad_out, idx, N, need_xyz=True, need_center=True):
    _need(grad_out, "grad_out", torch.float32, 4, 3)
    _need(idx, "idx", torch.int64, 3)
    B, G, K = idx.shape
    gx = torch.zeros((B, N, 3), dtype=torch.float32, device=grad_out.device) if need_xyz else None
    gc = torch.empty((B, G, 3), dtype=torch.float32, device=grad_out.device) if need_center else None
    _call(grad_out.device, "upp_group_bwd", _abi.ptr(grad_out), _abi.ptr(idx), _abi.ptr(gx), _abi.ptr(gc), B, N, G, K)
    return gx, gc


def fps_gather_bwd(grad_centers, idx, N):
    """Gradient of the FPS centre gather: grad_centers (B,M,3), idx (B,M) int32 -> (B,N,3) (upp_fps_gather_bwd: one launch)."""
    _need(grad_centers, "grad_centers", torch.float32, 3, 3)
    _need(idx, "idx", torch.int32, 2)
    B, M = idx.shape
    if tuple(grad_centers.shape) != (B, M, 3):
        raise RuntimeError("fps_gather_bwd: grad_centers must be (B, M, 3) for idx (B, M)")
    gx = torch.empty((B, int(N), 3), dtype=torch.float32, device=grad_centers.device)
    if B:
        _call(grad_centers.device, "upp_fps_gather_bwd", _abi.ptr(grad_centers), _abi.ptr(idx), _abi.ptr(gx), B, int(N), M)
    return gx


# ------------------------------------------------------------------ Chamfer
def chamfer_fwd(xyz1, xyz2):
    _need(xyz1, "xyz1", torch.float32, 3, 3)
    _need(xyz2, "xyz2", torch.float32, 3, 3)
    _same_device(xyz1, xyz2)
    B, n, _ = xyz1.shape
    if xyz2.shape[0] != B:
        raise RuntimeError("xyz1 and xyz2 batch sizes differ")
    m = xyz2.shape[1]
    dev = xyz1.device
    dist1 = torch.empty((B, n), dtype=torch.float32, device=dev)
    dist2 = torch.empty((B, m), dtype=torch.float32, device=dev)
    idx1 = torch.empty((B, n), dtype=torch.int32, device=dev)
    idx2 = torch.empty((B, m), dtype=torch.int32, device=dev)
    if B == 0:
        return dist1, dist2, idx1, idx2
    _call(xyz1.device, "upp_chamfer_fwd", _abi.ptr(xyz1), _abi.ptr(xyz2), _abi.ptr(dist1), _abi.ptr(dist2),
          _abi.ptr(idx1), _abi.ptr(idx2), B, n, m)
    return dist1, dist2, idx1, idx2


def chamfer_bwd(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2):
    _need(xyz1, "xyz1", torch.float32, 3, 3)
    _need(xyz2, "xyz2", torch.float32, 3, 3)
    _need(idx1, "idx1", torch.int32, 2)
    _need(idx2, "idx2", torch.int32, 2)
    grad_dist1 = grad_dist1.contiguous()
    grad_dist2 = grad_dist2.contiguous()
    _need(grad_dist1, "grad_dist1", torch.float32, 2)
    _need(grad_dist2, "grad_dist2", torch.float32, 2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = torch.empty_like(xyz1)          # (overwritten by the kernel: include/upp_hip.h)
    g2 = torch.empty_like(xyz2)
    if B == 0:
        return g1, g2
    _call(xyz1.device, "upp_chamfer_bwd", _abi.ptr(xyz1), _abi.ptr(xyz2), _abi.ptr(idx1), _abi.ptr(idx2),
          _abi.ptr(grad_dist1), _abi.ptr(grad_dist2), _abi.ptr(g1), _abi.ptr(g2), B, n, m)
    return g1, g2


def chamfer_loss(dist1, dist2, l1):
    """-> (loss (1,), fac1 (B,n), fac2 (B,m)): the reduction of ChamferDistanceL1 (l1) / L2 and d loss / d dist (upp_chamfer_loss)."""
    _need(dist1, "dist1", torch.float32, 2)
    _need(dist2, "dist2", torch.float32, 2)
    B, n = dist1.shape
    m = dist2.shape[1]
    dev = dist1.device
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    fac1, fac2 = torch.empty_like(dist1), torch.empty_like(dist2)
    work = torch.empty((int(_abi.load().upp_chamfer_loss_work_floats()),), dtype=torch.float32, device=dev)
    _call(dev, "upp_chamfer_loss", _abi.ptr(dist1), _abi.ptr(dist2), B, n, m, 1 if l1 else 0, _abi.ptr(loss), _abi.ptr(fac1), _abi.ptr(fac2), _abi.ptr(work))
    return loss, fac1, fac2


# ------------------------------------------------------------------ EMD
def emd_approxmatch(xyz1, xyz2):
    _need(xyz1, "xyz1", torch.float32, 3, 3)
    _need(xyz2, "xyz2", torch.float32, 3, 3)
    _same_device(xyz1, xyz2)
    B, n, _ = xyz1.shape
    if xyz2.shape[0] != B:
        raise RuntimeError("xyz1 and xyz2 batch sizes differ")
    m = xyz2.shape[1]
    match = torch.empty((B, m, n), dtype=torch.float32, device=xyz1.device)
    if B == 0:
        return match
    nwork = int(_abi.load().upp_emd_work_floats(B, n, m))
    work = torch.empty((max(nwork, 1),), dtype=torch.float32, device=xyz1.device)
    _call(xyz1.device, "upp_emd_approxmatch", _abi.ptr(xyz1), _abi.ptr(xyz2), _abi.ptr(match), _abi.ptr(work), B, n, m)
    return match


def emd_matchcost(xyz1, xyz2, match):
    _need(xyz1, "xyz1", torch.float32, 3, 3)
    _need(xyz2, "xyz2", torch.float32, 3, 3)
    _need(match, "match", torch.float32, 3)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    if tuple(match.shape) != (B, m, n):
        raise RuntimeError("match must be (B, n2, n1)")
    cost = torch.empty((B,), dtype=torch.float32, device=xyz1.device)
    if B == 0:
        return cost
    _call(xyz1.device, "upp_emd_matchcost", _abi.ptr(xyz1), _abi.ptr(xyz2), _abi.ptr(match), _abi.ptr(cost), B, n, m)
    return cost


def emd_matchcost_bwd(grad_cost, xyz1, xyz2, match):
    _need(grad_cost, "grad_cost", torch.float32, 1)
    _need(xyz1, "xyz1", torch.float32, 3, 3)
    _need(xyz2, "xyz2", torch.float32, 3, 3)
    _need(match, "match", torch.float32, 3)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    if B == 0:
        return g1, g2
    _call(xyz1.device, "upp_emd_matchcost_bwd", _abi.ptr(grad_cost), _abi.ptr(xyz1), _abi.ptr(xyz2), _abi.ptr(match),
          _abi.ptr(g1), _abi.ptr(g2), B, n, m)
    return g1, g2


# ------------------------------------------------------------------ patch embedding (forward only)
def patch_embed_fwd(point_groups, enc, training):
    """point_groups (B,G,n,3) -> (B,G,C) through upp_patch_embed_fwd.  `enc` is an Encoder-shaped
    module (first_conv / second_conv Sequentials with the reference's layer sizes)."""
    _need(point_groups, "point_groups", torch.float32, 4, 3)
    B, G, n, _ = point_groups.shape
    c1, bn1, _, c2 = enc.first_conv
    c3, bn3, _, c4 = enc.second_conv
    C = c4.weight.shape[0]
    R = B * G * n
    dev = point_groups.device
    work = torch.empty((int(_abi.load().upp_patch_embed_work_floats(R, n)),), dtype=torch.float32, device=dev)
    out = torch.empty((B, G, C), dtype=torch.float32, device=dev)
    mom = 0.1 if bn1.momentum is None else float(bn1.momentum)
    _call(dev, "upp_patch_embed_fwd", _abi.ptr(point_groups), R, n,
          _abi.ptr(c1.weight), _abi.ptr(c1.bias), _abi.ptr(bn1.weight), _abi.ptr(bn1.bias),
          _abi.ptr(bn1.running_mean), _abi.ptr(bn1.running_var),
          _abi.ptr(c2.weight), _abi.ptr(c2.bias),
          _abi.ptr(c3.weight), _abi.ptr(c3.bias), _abi.ptr(bn3.weight), _abi.ptr(bn3.bias),
          _abi.ptr(bn3.running_mean), _abi.ptr(bn3.running_var),
          _abi.ptr(c4.weight), _abi.ptr(c4.bias), C, mom, float(bn1.eps), 1 if training else 0,
          _abi.ptr(work), _abi.ptr(out))
    return out


# ------------------------------------------------------------------ Transformer block glue
LIN_NONE, LIN_BIAS, LIN_BIAS_GELU, LIN_BIAS_GELU_D, LIN_MUL, LIN_BIAS_RELU = 0, 1, 2, 3, 4, 5


class time_linear_calls:
    """Measurement scope (bench.py): every upp_linear_f32 / upp_linear_sb_f32 launch inside it is bracketed by a pair of HIP events recorded
    on the launch stream; `.report()` -> [(M, N, K, epilogue, ms, sb)] after a synchronize (sb: the split-bf16 kernel's tile code, 0 = the
    exact-f32 kernel).  Eager launches only (not under capture)."""
    active = None

    def __enter__(self):
        self.calls = []
        self.wgrad_groups = []          # [(M, N, K), ...] per upp_linear_wgrad_grouped_f32 launch
        time_linear_calls.active = self
        return self

    def __exit__(self, *exc):
        time_linear_calls.active = None
        return False

    def report(self):
        torch.cuda.synchronize()
        return [(M, N, K, e, a.elapsed_time(b), sb) for (M, N, K, e, a, b, sb) in self.calls]


SPLIT_BF16 = os.environ.get("UPP_SPLIT_BF16", "1") != "0"      # frozen weights on the bf16 matrix pipe at f32 accuracy (csrc/linear_sb.hip)


class _WeightPlanes:
    """bf16 plane images (upp_linear_sb_prep) of FROZEN weights and of their cached transposes.  Cached -- one image per weight, made on
    first use (the eager warm-up of a captured step) and refreshed IN PLACE when the version counter of the weight's storage moved
    (load_state_dict, the in-place refresh of a cached W^T), so that HIP graphs that captured the image's address stay valid -- only for
    weights whose storage owner is PERSISTENT: an nn.Parameter, or a buffer marked `_upp_persistent` by its keeper (functional.TRANSPOSED's
    W^T copies, the patch embedding's zero-padded first-layer weight).  An entry is valid for the owner OBJECT it was made for: another
    tensor that lands on the same address gets a new image.  Any other weight (a `.contiguous()` copy of a misaligned column window, a
    padded temporary) is split afresh at every use into a buffer of its own: replacing a cached image would free memory that an already
    captured graph still reads.  `refresh()` re-splits every live entry (functional.refresh_caches: weights loaded into a model whose
    step is already captured).
    CONTRACT: validity is keyed on torch's version counter of the owner.  A write that bypasses it -- `p.data.copy_()`, `p.data.mul_()`, a
    custom kernel writing a frozen weight -- leaves the image stale without any error: call functional.refresh_caches(model) after such
    a write (README "Changing frozen weights").  train.TrainStep calls it at construction."""

    def __init__(self):
        self.entries = {}
        self.trainable = {}

    @staticmethod
    def _split(w, planes=None, transposed=False):
        """planes of the (N,K) operand `w`, or -- transposed -- of w^T for a row-major w (K,N) (upp_linear_sb_prep reads it transposed)."""
        N, K = (w.shape[1], w.shape[0]) if transposed else w.shape
        if planes is None:
            planes = torch.empty(int(_abi.load().upp_linear_sb_planes_bytes(N, K)), dtype=torch.uint8, device=w.device)
        _call(w.device, "upp_linear_sb_prep", _abi.ptr(w), w.stride(0), N, K, 1 if transposed else 0, _abi.ptr(planes))
        return planes

    # -- trainable weights inside a step driver (train.TrainStep sets `managed`): the plane images of W (forward) and of W^T (data gradient,
    #    split straight from W) are persistent and ALL re-split by one launch at the start of every step (`refresh_trainable`), i.e. after
    #    whatever changed the weights since the last step (the flat AdamW kernel writes them without touching torch's version counters).
    #    Outside a step driver a trainable weight stays on the exact-f32 kernel (it may have changed by any means).
    managed = False

    def get_trainable(self, w, transposed=False):
        owner = w._base if w._base is not None else w
        if not isinstance(owner, torch.nn.Parameter):           # a computed weight (padded, concatenated ...): split where it is used
            return self._split(w.detach(), None, transposed)
        key = (w.data_ptr(), tuple(w.shape), tuple(w.stride()), bool(transposed))
        e = self.trainable.get(key)
        if e is None or e[0]() is not owner:
            planes = self._split(w.detach(), None, transposed)
            self.trainable[key] = [weakref.ref(owner), planes, bool(transposed), (tuple(w.shape), tuple(w.stride()), w.storage_offset())]
            return planes
        return e[1]

    def refresh_trainable(self):
        import ctypes
        jobs = []
        for key, e in list(self.trainable.items()):
            owner = e[0]()
            if owner is None:
                del self.trainable[key]
            else:
                jobs.append((torch.as_strided(owner.detach(), e[3][0], e[3][1], e[3][2]), e[1], e[2]))
        if not jobs:
            return
        k = len(jobs)
        W = (ctypes.c_void_p * k)(*[w.data_ptr() for w, _, _ in jobs])
        ld = (ctypes.c_longlong * k)(*[w.stride(0) for w, _, _ in jobs])
        N = (ctypes.c_int * k)(*[(w.shape[1] if tr else w.shape[0]) for w, _, tr in jobs])
        K = (ctypes.c_int * k)(*[(w.shape[0] if tr else w.shape[1]) for w, _, tr in jobs])
        T = (ctypes.c_int * k)(*[1 if tr else 0 for _, _, tr in jobs])
        P = (ctypes.c_void_p * k)(*[p.data_ptr() for _, p, _ in jobs])
        _call(jobs[0][0].device, "upp_linear_sb_prep_batched", W, ld, N, K, T, P, k)

    def get(self, w):
        owner = w._base if w._base is not None else w
        if not (isinstance(owner, torch.nn.Parameter) or getattr(owner, "_upp_persistent", False)):
            return self._split(w)
        key = (w.data_ptr(), tuple(w.shape), tuple(w.stride()))
        e = self.entries.get(key)
        if e is None or e[0]() is not owner:
            if len(self.entries) > 1024:
                self.entries = {k: v for k, v in self.entries.items() if v[0]() is not None}
            planes = self._split(w)
            self.entries[key] = [weakref.ref(owner), owner._version, planes, (tuple(w.shape), tuple(w.stride()), w.storage_offset())]
            return planes
        if e[1] != owner._version:
            if torch.cuda.is_current_stream_capturing():
                # the image is stale and a re-split here would be baked into the graph as a launch of its own: refuse loudly instead of
                # multiplying by the old weight (round-4 advisor).  Weights change BETWEEN captures: refresh the caches there.
                raise RuntimeError("upp_hip: a frozen weight changed after its bf16 plane image was made and the stream is capturing; call "
                                   "upp_hip.functional.refresh_caches(model) after changing weights and before capturing a step")
            self._split(w, e[2])
            e[1] = owner._version
        return e[2]

    def refresh(self):
        for e in self.entries.values():
            owner = e[0]()
            if owner is not None:
                self._split(torch.as_strided(owner.detach(), e[3][0], e[3][1], e[3][2]), e[2])
                e[1] = owner._version


PLANES = _WeightPlanes()


def linear_sb_tile(M, N, K):
    """Tile code of the split-bf16 kernel for (M,N,K), 0 = not a problem for it (upp_linear_sb_tile)."""
    return max(0, int(_abi.load().upp_linear_sb_tile(int(M), int(N), int(K))))


def _sb_servable(N, K, out, bias, aux):
    return (N % 4 == 0 and K % 32 == 0 and out.data_ptr() % 16 == 0 and (bias is None or bias.data_ptr() % 16 == 0)
            and (aux is None or aux.data_ptr() % 16 == 0))


def linear_sb_usable(M, N, K):
    """Does the split-bf16 kernel take an (M,K) x (N,K)^T product (freshly allocated, 16-byte aligned operands assumed)?"""
    return bool(SPLIT_BF16 and N % 4 == 0 and K % 32 == 0 and linear_sb_tile(M, N, K))


def linear_f32(a, w, bias=None, epilogue=LIN_NONE, aux=None, tile=0, out=None, frozen=False, planes=None, wshape=None):
    """C (M,N) = epilogue(a (M,K) . w (N,K)^T) on the matrix cores: upp_linear_f32 (exact f32 MFMA, bit-pinned order), or -- frozen=True: w
    is a frozen weight or a cached copy of one, its contents change only through torch (version counter) -- upp_linear_sb_f32 (three-way
    bf16 split of both operands, six bf16 MFMA products, f32 accumulate: f32 accuracy at 6/16 of the matrix-pipe time) where that kernel
    takes the shape.  planes / wshape: the plane image of the (N,K) operand made by the caller (ops.PLANES.get_trainable: a trainable
    weight inside a step driver, or its transpose) -- then w may be None and the split-bf16 kernel is taken (linear_sb_usable must hold).
    a: (..., K) f32 whose rows are K-contiguous with one common row stride; w: (N,K).
    epilogue LIN_BIAS_GELU_D returns (C, GELU'); LIN_MUL multiplies by aux (M,N)."""
    if planes is not None:
        N, Kw = wshape
    else:
        if not (isinstance(w, torch.Tensor) and w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.stride(1) == 1):
            raise RuntimeError("w must be a 2-D f32 HIP (cuda) matrix with contiguous rows; upp_hip has no CPU path")
        _same_device(a, w)
        N, Kw = w.shape
    if not (isinstance(a, torch.Tensor) and a.is_cuda and a.dtype == torch.float32):
        raise RuntimeError("a must be a f32 HIP (cuda) tensor; upp_hip has no CPU path")
    K = a.shape[-1]
    if Kw != K:
        raise RuntimeError(f"linear_f32: a (...,{K}) against w ({N},{Kw})")
    a2 = a.reshape(-1, K)
    if a2.stride(1) != 1 or a2.stride(0) % 4 != 0 or a2.data_ptr() % 16 != 0:      # (a column window with a misaligned base: one copy)
        a2 = a2.contiguous()
    M = a2.shape[0]
    lead = tuple(a.shape[:-1])
    if out is None:
        out = torch.empty(lead + (N,), dtype=torch.float32, device=a.device)
    d = None
    if epilogue == LIN_BIAS_GELU_D:
        d = torch.empty(lead + (N,), dtype=torch.float32, device=a.device)
        aux = d
    if epilogue == LIN_MUL:
        _need(aux, "aux", torch.float32)
        if aux.numel() != M * N:
            raise RuntimeError("linear_f32: aux must be (M,N)")
    if bias is not None:
        _need(bias, "bias", torch.float32, 1, N)
    sb = 0
    if planes is None and frozen and SPLIT_BF16 and tile == 0 and _sb_servable(N, K, out, bias, aux):
        sb = linear_sb_tile(M, N, K)
        if sb:
            planes = PLANES.get(w)
    elif planes is not None:
        sb = linear_sb_tile(M, N, K)
        if not (sb and _sb_servable(N, K, out, bias, aux)):
            raise RuntimeError("linear_f32: plane images given for a problem the split-bf16 kernel does not take (ask linear_sb_usable first)")
    scope = time_linear_calls.active
    if scope is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if sb:
        _call(a.device, "upp_linear_sb_f32", _abi.ptr(a2), a2.stride(0), _abi.ptr(planes), _abi.ptr(bias), _abi.ptr(out), N,
              _abi.ptr(aux), N, M, N, K, int(epilogue), sb)
    else:
        _call(a.device, "upp_linear_f32", _abi.ptr(a2), a2.stride(0), _abi.ptr(w), w.stride(0), _abi.ptr(bias), _abi.ptr(out), N,
              _abi.ptr(aux), N, M, N, K, int(epilogue), int(tile))
    if scope is not None:
        ev1.record()
        scope.calls.append((M, N, K, int(epilogue), ev0, ev1, sb))
    return (out, d) if epilogue == LIN_BIAS_GELU_D else out


def colsum_partials(part, offset, length, chunks=None):
    """(chunks, length) partial column sums of columns [offset, offset + length) of the tall 2-D matrix `part` (upp_colsum_partials)."""
    if not (isinstance(part, torch.Tensor) and part.is_cuda and part.dtype == torch.float32 and part.dim() == 2 and part.stride(1) == 1):
        raise RuntimeError("colsum_partials: a 2-D f32 HIP (cuda) matrix with contiguous rows is required; upp_hip has no CPU path")
    if offset < 0 or length < 1 or offset + length > part.shape[1]:
        raise RuntimeError(f"colsum_partials: columns [{offset}, {offset + length}) outside {tuple(part.shape)}")
    n = part.shape[0]
    if chunks is None:
        chunks = max(1, min(256, (n + 255) // 256))
    dst = torch.empty((chunks, length), dtype=torch.float32, device=part.device)
    view = part[:, offset:offset + length]
    _call(part.device, "upp_colsum_partials", _abi.ptr(view), part.stride(0), n, length, chunks, _abi.ptr(dst))
    return dst


def sum_rows(part, offset=0, length=None):
    """Column sums of columns [offset, offset + length) of a 2-D f32 HIP matrix on this library's kernels (two stages of
    upp_colsum_partials for tall matrices) -> (length,).  NOT torch.sum: a torch reduction that splits its rows over workgroups zeroes its
    semaphores with a memset, and a memset NODE of a captured step graph works in the first replay only on this stack (NOTEBOOK 12.11)."""
    if length is None:
        length = part.shape[1] - offset
    while part.shape[0] > 256:
        part, offset = colsum_partials(part, offset, length), 0
    return colsum_partials(part, offset, length, chunks=1)[0]


def logsoftmax_rows_fwd(y, bias, C):
    """log_softmax over the first C columns of the rows of y (R, >= C) (+ bias (C)) -> logp (R, C); upp_logsoftmax_rows_fwd."""
    if not (isinstance(y, torch.Tensor) and y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.stride(1) == 1 and y.shape[1] >= C):
        raise RuntimeError("logsoftmax_rows_fwd: a 2-D f32 HIP (cuda) matrix with contiguous rows of at least C columns is required; upp_hip has no CPU path")
    if bias is not None:
        _need(bias, "bias", torch.float32, 1, int(C))
    R = y.shape[0]
    logp = torch.empty((R, int(C)), dtype=torch.float32, device=y.device)
    _call(y.device, "upp_logsoftmax_rows_fwd", _abi.ptr(y), y.stride(0), _abi.ptr(bias), R, int(C), _abi.ptr(logp))
    return logp


def logsoftmax_rows_bwd(g_logp, logp, Cpad):
    """-> g_y (R, Cpad): the log-softmax backward in the first C columns, zeros in the pad columns; upp_logsoftmax_rows_bwd."""
    _need(g_logp, "g_logp", torch.float32, ndim=2)
    _need(logp, "logp", torch.float32, ndim=2)
    R, C = logp.shape
    if tuple(g_logp.shape) != (R, C):
        raise RuntimeError("logsoftmax_rows_bwd: g_logp must match logp")
    g_y = torch.empty((R, int(Cpad)), dtype=torch.float32, device=logp.device)
    _call(logp.device, "upp_logsoftmax_rows_bwd", _abi.ptr(g_logp), _abi.ptr(logp), R, C, _abi.ptr(g_y), int(Cpad), int(Cpad))
    return g_y


def nll_mean_fwd(logp, target):
    """F.nll_loss(logp, target) (mean) -> (1,) f32; upp_nll_mean_fwd (two launches, fixed-order sums)."""
    _need(logp, "logp", torch.float32, ndim=2)
    R, C = logp.shape
    if not (isinstance(target, torch.Tensor) and target.is_cuda and target.dtype == torch.int64 and target.numel() == R and target.is_contiguous()):
        raise RuntimeError("nll_mean_fwd: target must be a contiguous int64 HIP (cuda) tensor with one entry per row")
    part = torch.empty(int(_abi.load().upp_nll_mean_part_floats(R)), dtype=torch.float32, device=logp.device)
    out = torch.empty(1, dtype=torch.float32, device=logp.device)
    _call(logp.device, "upp_nll_mean_fwd", _abi.ptr(logp), _abi.ptr(target), R, C, _abi.ptr(part), _abi.ptr(out))
    return out


def nll_mean_bwd(g_loss, target, R, C):
    """-> g_logp (R, C) = -g_loss / R at (r, target[r]), 0 elsewhere; g_loss a one-element device tensor; upp_nll_mean_bwd."""
    _need(g_loss, "g_loss", torch.float32)
    g_logp = torch.empty((int(R), int(C)), dtype=torch.float32, device=target.device)
    _call(target.device, "upp_nll_mean_bwd", _abi.ptr(g_loss), _abi.ptr(target), int(R), int(C), _abi.ptr(g_logp))
    return g_logp


def noise_loss_fwd(pred, noise_vector, pn, want_score=True):
    """-> (loss (1,), score (B,P) or None) of the pre-task noise supervision (upp_noise_loss_fwd): pred (B,P,3), noise_vector (B,P-pn,3)."""
    _need(pred, "pred", torch.float32, 3, 3)
    _need(noise_vector, "noise_vector", torch.float32, 3, 3)
    B, P, _ = pred.shape
    if tuple(noise_vector.shape) != (B, P - int(pn), 3):
        raise RuntimeError("noise_loss_fwd: noise_vector must be (B, P - pn, 3)")
    part = torch.empty(int(_abi.load().upp_noise_loss_part_floats(B, P)), dtype=torch.float32, device=pred.device)
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    score = torch.empty((B, P), dtype=torch.float32, device=pred.device) if want_score else None
    _call(pred.device, "upp_noise_loss_fwd", _abi.ptr(pred), _abi.ptr(noise_vector), B, P, int(pn), _abi.ptr(part), _abi.ptr(loss), _abi.ptr(score))
    return loss, score


def noise_loss_bwd(g_loss, pred, noise_vector, pn):
    _need(g_loss, "g_loss", torch.float32)
    B, P, _ = pred.shape
    g_pred = torch.empty_like(pred)
    _call(pred.device, "upp_noise_loss_bwd", _abi.ptr(g_loss), _abi.ptr(pred), _abi.ptr(noise_vector), B, P, int(pn), _abi.ptr(g_pred))
    return g_pred


def rectify_select(feature, w0, b0, w1, b1, pts, keep, u=None, p=0.0, factor=1.0, nudge=0.2, want_pred=False, want_order=False, want_score=False):
    """Tail of the denoising prompter in two launches (upp_rectify_select): pred = score head(feature) * factor, moved = pts + nudge * pred,
    the `keep` least suspicious points of every cloud in descending-score order.  -> out (B,keep,3) [, pred (B,N,3)] [, order (B,N) int64] [, score (B,N)]."""
    _need(pts, "pts", torch.float32, 3, 3)
    B, N, _ = pts.shape
    for t, n_, shp in ((feature, "feature", (B, N, 32)), (w0, "w0", (64, 32)), (b0, "b0", (64,)), (w1, "w1", (3, 64)), (b1, "b1", (3,))):
        _need(t, n_, torch.float32)
        if tuple(t.shape) != shp:
            raise RuntimeError(f"rectify_select: {n_} must be {shp}, got {tuple(t.shape)}")
    if u is not None:
        _need(u, "u", torch.float32)
        if u.numel() != B * N * 64:
            raise RuntimeError("rectify_select: u must hold B*N*64 uniforms")
    dev = pts.device
    moved = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
    score = torch.empty((B, N), dtype=torch.float32, device=dev)
    out = torch.empty((B, int(keep), 3), dtype=torch.float32, device=dev)
    pred = torch.empty((B, N, 3), dtype=torch.float32, device=dev) if want_pred else None
    order = torch.empty((B, N), dtype=torch.int64, device=dev) if want_order else None
    _call(dev, "upp_rectify_select", _abi.ptr(feature), _abi.ptr(w0), _abi.ptr(b0), _abi.ptr(w1), _abi.ptr(b1), _abi.ptr(u), float(p), float(factor),
          _abi.ptr(pts), float(nudge), B, N, int(keep), _abi.ptr(pred), _abi.ptr(moved), _abi.ptr(score), _abi.ptr(out), _abi.ptr(order))
    res = (out,) + ((pred,) if want_pred else ()) + ((order,) if want_order else ()) + ((score,) if want_score else ())
    return res if len(res) > 1 else out


def group_max_fwd(x):
    """x (R,k,C) f32 contiguous -> (max over k (R,C), arg-max (R,C) uint8)  (upp_group_max_fwd)."""
    _need(x, "x", torch.float32, ndim=3)
    R, k, C = x.shape
    out = torch.empty((R, C), dtype=torch.float32, device=x.device)
    amax = torch.empty((R, C), dtype=torch.uint8, device=x.device)
    _call(x.device, "upp_group_max_fwd", _abi.ptr(x), R, k, C, _abi.ptr(out), _abi.ptr(amax))
    return out, amax


def group_max_bwd(g, amax, k):
    """g (R,C), amax (R,C) uint8 -> g_x (R,k,C): g at the arg-max row of every (group, column), zero elsewhere (upp_group_max_bwd)."""
    _need(g, "g", torch.float32, ndim=2)
    R, C = g.shape
    g_x = torch.empty((R, int(k), C), dtype=torch.float32, device=g.device)
    _call(g.device, "upp_group_max_bwd", _abi.ptr(g), _abi.ptr(amax), R, int(k), C, _abi.ptr(g_x))
    return g_x


def argsort_rows(key, descending=False):
    """Stable argsort of every row of `key` (..., N) f32 -> int64 indices of the same shape (upp_argsort_rows: rank counting, no library
    sort; NaN ranks as +inf, equal keys in index order)."""
    _need(key, "key", torch.float32)
    N = key.shape[-1]
    k2 = key.reshape(-1, N).contiguous()
    order = torch.empty(k2.shape, dtype=torch.int64, device=key.device)
    _call(key.device, "upp_argsort_rows", _abi.ptr(k2), k2.shape[0], N, 1 if descending else 0, _abi.ptr(order))
    return order.view(key.shape)


def wcolsum_partials(src, wts, chunks=None):
    """(chunks, W, C) partials of wts (n,W)^T . src (n,C) for W <= 4 over very many rows (upp_wcolsum_partials); sum over dim 0 = the product."""
    for t_, n_ in ((src, "src"), (wts, "wts")):
        if not (isinstance(t_, torch.Tensor) and t_.is_cuda and t_.dtype == torch.float32 and t_.dim() == 2 and t_.stride(1) == 1):
            raise RuntimeError(f"{n_} must be a 2-D f32 HIP (cuda) matrix with contiguous rows; upp_hip has no CPU path")
    _same_device(src, wts)
    n, C = src.shape
    W = wts.shape[1]
    if wts.shape[0] != n or W > 4 or C % 4:
        raise RuntimeError(f"wcolsum_partials: wts {tuple(wts.shape)} against src {tuple(src.shape)} (W <= 4, C % 4 == 0)")
    if chunks is None:
        chunks = max(1, min(256, (n + 255) // 256))
    dst = torch.empty((chunks, W, C), dtype=torch.float32, device=src.device)
    _call(src.device, "upp_wcolsum_partials", _abi.ptr(src), src.stride(0), _abi.ptr(wts), wts.stride(0), W, n, C, chunks, _abi.ptr(dst))
    return dst


def linear_smallk(x, w, bias=None, act=0):
    """act(x (...,K) . w (N,K)^T + bias) for K <= 64, N <= 256, any alignment (upp_linear_smallk_f32); act 0 none / 1 ReLU / 2 GELU."""
    for t_, n_ in ((x, "x"), (w, "w")):
        if not (isinstance(t_, torch.Tensor) and t_.is_cuda and t_.dtype == torch.float32):
            raise RuntimeError(f"{n_} must be a f32 HIP (cuda) tensor; upp_hip has no CPU path")
    _same_device(x, w)
    K, N = x.shape[-1], w.shape[0]
    if w.dim() != 2 or w.shape[1] != K:
        raise RuntimeError(f"linear_smallk: x (...,{K}) against w {tuple(w.shape)}")
    if bias is not None:
        _need(bias, "bias", torch.float32, 1, N)
    if K > 64 or N > 256:
        raise RuntimeError(f"linear_smallk serves K <= 64 and N <= 256, got K = {K}, N = {N}")
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    if w.stride(1) != 1:
        w = w.contiguous()
    out = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float32, device=x.device)
    _call(x.device, "upp_linear_smallk_f32", _abi.ptr(x2), x2.stride(0), _abi.ptr(w), w.stride(0), _abi.ptr(bias), _abi.ptr(out), N,
          x2.shape[0], N, K, int(act))
    return out


def linear_smallk_wgrad(g, x, chunks=None):
    """(chunks, N, K) partial weight gradients of a small Linear: sum over dim 0 = g^T . x (g (M,N), x (M,K), N K <= 2048);
    upp_linear_smallk_wgrad_f32."""
    for t_, n_ in ((g, "g"), (x, "x")):
        if not (isinstance(t_, torch.Tensor) and t_.is_cuda and t_.dtype == torch.float32 and t_.dim() == 2 and t_.stride(1) == 1):
            raise RuntimeError(f"{n_} must be a 2-D f32 HIP (cuda) matrix with contiguous rows; upp_hip has no CPU path")
    M, N = g.shape
    K = x.shape[1]
    if x.shape[0] != M or N * K > 2048:
        raise RuntimeError(f"linear_smallk_wgrad: g {tuple(g.shape)} against x {tuple(x.shape)} (N K <= 2048)")
    if chunks is None:
        chunks = max(1, min(512, (M + 63) // 64))
    part = torch.empty((chunks, N, K), dtype=torch.float32, device=g.device)
    _call(g.device, "upp_linear_smallk_wgrad_f32", _abi.ptr(g), g.stride(0), _abi.ptr(x), x.stride(0), _abi.ptr(part), M, N, K, chunks)
    return part


def linear_smallk_gelu_d(x, w, bias):
    """-> (GELU(x . w^T + bias), GELU'(x . w^T + bias)) for K <= 64, N <= 256 in one pass (upp_linear_smallk_gelu_d_f32)."""
    for t_, n_ in ((x, "x"), (w, "w")):
        if not (isinstance(t_, torch.Tensor) and t_.is_cuda and t_.dtype == torch.float32):
            raise RuntimeError(f"{n_} must be a f32 HIP (cuda) tensor; upp_hip has no CPU path")
    _same_device(x, w)
    K, N = x.shape[-1], w.shape[0]
    if w.dim() != 2 or w.shape[1] != K or K > 64 or N > 256:
        raise RuntimeError(f"linear_smallk_gelu_d: x (...,{K}) against w {tuple(w.shape)} (K <= 64, N <= 256)")
    if bias is not None:
        _need(bias, "bias", torch.float32, 1, N)
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    if w.stride(1) != 1:
        w = w.contiguous()
    out = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float32, device=x.device)
    d = torch.empty_like(out)
    _call(x.device, "upp_linear_smallk_gelu_d_f32", _abi.ptr(x2), x2.stride(0), _abi.ptr(w), w.stride(0), _abi.ptr(bias), _abi.ptr(out), N,
          _abi.ptr(d), N, x2.shape[0], N, K)
    return out, d


def transpose(w, out=None):
    """W^T of a 2-D f32 matrix with contiguous rows (upp_transpose_f32); `out` (cols, rows) is overwritten in place when given."""
    if not (isinstance(w, torch.Tensor) and w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.stride(1) == 1):
        raise RuntimeError("transpose: a 2-D f32 HIP (cuda) matrix with contiguous rows is required (column windows are fine)")
    rows, cols = w.shape
    if out is None:
        out = torch.empty((cols, rows), dtype=torch.float32, device=w.device)
    _call(w.device, "upp_transpose_f32", _abi.ptr(w), w.stride(0), _abi.ptr(out), out.stride(0), rows, cols)
    return out


def linear_wgrad(g, x):
    """Partial weight gradients (splits, N, K) of y = x . W^T: sum over dim 0 = g^T . x  (g (M,N), x (M,K); upp_linear_wgrad_f32).
    The caller sums the splits (batched_sum / the deferred sums of a training step)."""
    for t, name in ((g, "g"), (x, "x")):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise RuntimeError(f"{name} must be a 2-D f32 HIP (cuda) matrix with contiguous rows; upp_hip has no CPU path")
    _same_device(g, x)
    M, N = g.shape
    K = x.shape[1]
    if x.shape[0] != M:
        raise RuntimeError("linear_wgrad: row counts differ")
    if SPLIT_BF16 and WGRAD_SPLIT_BF16:
        return linear_wgrad_grouped([(g, x)])[0]
    splits = _abi.load().upp_linear_wgrad_splits(M, N, K)
    part = torch.empty((splits, N, K), dtype=torch.float32, device=g.device)
    _call(g.device, "upp_linear_wgrad_f32", _abi.ptr(g), g.stride(0), _abi.ptr(x), x.stride(0), _abi.ptr(part), M, N, K)
    return part


def linear_group_bias_usable(M, N, K, rows_per_group):
    """Does upp_linear_group_bias_f32 serve this problem (tall: the register-tiled kernel; groups of 2^s >= 32 rows; N, K % 4 == 0)?"""
    r = int(rows_per_group)
    if not (r >= 32 and (r & (r - 1)) == 0 and M % r == 0 and N % 4 == 0 and K % 4 == 0):
        return False
    tile = int(_abi.load().upp_linear_tile(M, N, K))
    return tile > 0 and bool(tile & 0x10000)


def linear_group_bias(a, w, bias, rows_per_group, frozen=False, planes=None):
    """C (M,N) = a (M,K) . w (N,K)^T + bias[m // rows_per_group] -- the bias of a GROUP of rows added in the GEMM's epilogue
    (upp_linear_group_bias_f32, or upp_linear_sb_group_bias_f32 for a frozen weight / a plane image handed over by the caller: see
    linear_f32); bias (M / rows_per_group, N) contiguous."""
    for t, name in ((a, "a"), (w, "w"), (bias, "bias")):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise RuntimeError(f"{name} must be a 2-D f32 HIP (cuda) matrix with contiguous rows; upp_hip has no CPU path")
    _same_device(a, w, bias)
    M, K = a.shape
    N = w.shape[0]
    r = int(rows_per_group)
    if w.shape[1] != K or not bias.is_contiguous() or tuple(bias.shape) != (M // r, N) or not linear_group_bias_usable(M, N, K, r):
        raise RuntimeError("linear_group_bias: shapes do not fit (see linear_group_bias_usable)")
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    sb = 0
    if (planes is not None or frozen) and linear_sb_usable(M, N, K) and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0 and bias.data_ptr() % 16 == 0:
        sb = linear_sb_tile(M, N, K)
        if planes is None:
            planes = PLANES.get(w)
    scope = time_linear_calls.active
    if scope is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if sb:
        _call(a.device, "upp_linear_sb_group_bias_f32", _abi.ptr(a), a.stride(0), _abi.ptr(planes), _abi.ptr(bias), r.bit_length() - 1,
              _abi.ptr(out), N, M, N, K)
    else:
        _call(a.device, "upp_linear_group_bias_f32", _abi.ptr(a), a.stride(0), _abi.ptr(w), w.stride(0), _abi.ptr(bias), r.bit_length() - 1,
              _abi.ptr(out), N, M, N, K)
    if scope is not None:
        ev1.record()
        scope.group_bias = getattr(scope, "group_bias", []) + [(M, N, K)]
        scope.calls.append((M, N, K, LIN_BIAS, ev0, ev1, sb))
    return out


WGRAD_SPLIT_BF16 = os.environ.get("UPP_WGRAD_SPLIT_BF16", "1") != "0"     # grouped weight gradients on the bf16 pipe (wgrad_sb.hip)


def linear_wgrad_grouped(pairs):
    """pairs: list of (g (M,N), x (M,K)) -- the weight gradients of several Linear layers in ONE launch (upp_linear_wgrad_grouped_f32).
    -> list of partial gradients (splits_p, N_p, K_p); the caller sums each over dim 0 in order (batched_sum)."""
    if not pairs:
        return []
    import ctypes
    k = len(pairs)
    dev = pairs[0][0].device
    for g, x in pairs:
        for t, name in ((g, "g"), (x, "x")):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.device == dev):
                raise RuntimeError(f"{name} must be a 2-D f32 matrix with contiguous rows on one HIP (cuda) device; upp_hip has no CPU path")
        if x.shape[0] != g.shape[0]:
            raise RuntimeError("linear_wgrad_grouped: row counts differ")
    M = (ctypes.c_int * k)(*[g.shape[0] for g, _ in pairs])
    N = (ctypes.c_int * k)(*[g.shape[1] for g, _ in pairs])
    K = (ctypes.c_int * k)(*[x.shape[1] for _, x in pairs])
    rows = (ctypes.c_int * k)()
    sb = SPLIT_BF16 and WGRAD_SPLIT_BF16
    _abi.check((_abi.load().upp_linear_wgrad_grouped_sb_rows if sb else _abi.load().upp_linear_wgrad_grouped_rows)(k, M, N, K, rows))
    parts = [torch.empty(((M[i] + rows[i] - 1) // rows[i], N[i], K[i]), dtype=torch.float32, device=dev) for i in range(k)]
    G = (ctypes.c_void_p * k)(*[g.data_ptr() for g, _ in pairs])
    X = (ctypes.c_void_p * k)(*[x.data_ptr() for _, x in pairs])
    P = (ctypes.c_void_p * k)(*[q.data_ptr() for q in parts])
    ldg = (ctypes.c_longlong * k)(*[g.stride(0) for g, _ in pairs])
    ldx = (ctypes.c_longlong * k)(*[x.stride(0) for _, x in pairs])
    _call(dev, "upp_linear_wgrad_grouped_sb" if sb else "upp_linear_wgrad_grouped_f32", G, ldg, X, ldx, P, M, N, K, rows, k)
    if time_linear_calls.active is not None:
        time_linear_calls.active.wgrad_groups.append([(M[i], N[i], K[i]) for i in range(k)])
    return parts


def rowln_fwd(x, add, prompts, mode, P, y, u, keep, gamma, beta, eps, Lout, want_xo=True, ybias=None):
    B, Lin, D = x.shape
    dev = x.device
    xo = torch.empty((B, Lout, D), dtype=torch.float32, device=dev) if want_xo else None
    if gamma is not None:
        h = torch.empty((B, Lout, D), dtype=torch.float32, device=dev)
        mean = torch.empty((B, Lout), dtype=torch.float32, device=dev)
        rstd = torch.empty((B, Lout), dtype=torch.float32, device=dev)
    else:
        h = mean = rstd = None
    _call(dev, "upp_rowln_fwd", _abi.ptr(x), _abi.ptr(add), _abi.ptr(prompts), int(mode), int(P), _abi.ptr(y), _abi.ptr(ybias),
          _abi.ptr(u), float(keep), _abi.ptr(gamma), _abi.ptr(beta), float(eps), _abi.ptr(xo), _abi.ptr(h), _abi.ptr(mean), _abi.ptr(rstd),
          B, Lin, Lout, D)
    return xo, h, mean, rstd


def rowln_bwd(g_xo, g_h, xo, mean, rstd, gamma, mode, u, keep, B, Lin, Lout, D, P, need_x, need_prompt, need_y, need_ln_part=False):
    dev = (g_xo if g_xo is not None else g_h).device
    g_x = torch.empty((B, Lin, D), dtype=torch.float32, device=dev) if need_x else None      # the kernel writes every row
    g_p = torch.empty((B, P, D), dtype=torch.float32, device=dev) if (need_prompt and P > 0 and mode in (1, 2)) else None
    g_y = torch.empty((B, Lin, D), dtype=torch.float32, device=dev) if need_y else None
    part = None
    if need_ln_part and g_h is not None:
        n = int(_abi.load().upp_rowln_part_floats(B, Lin, Lout, D, int(mode)))
        part = torch.empty((n // (2 * D), 2 * D), dtype=torch.float32, device=dev)     # per workgroup: [d_gamma | d_beta]
    _call(dev, "upp_rowln_bwd", _abi.ptr(g_xo), _abi.ptr(g_h), _abi.ptr(xo), _abi.ptr(mean), _abi.ptr(rstd), _abi.ptr(gamma),
          int(mode), _abi.ptr(u), float(keep), _abi.ptr(g_x), _abi.ptr(g_p), _abi.ptr(g_y), _abi.ptr(part), B, Lin, Lout, D, P)
    return g_x, g_p, g_y, part


def bias_gelu_fwd(z, bias):
    _need(z, "z", torch.float32)
    _need(bias, "bias", torch.float32, ndim=1, last=z.shape[-1])
    h = torch.empty_like(z)
    _call(z.device, "upp_bias_gelu_fwd", _abi.ptr(z), _abi.ptr(bias), _abi.ptr(h), z.numel() // z.shape[-1], z.shape[-1])
    return h


def bias_gelu_fwd_d(z, bias):
    """-> (GELU(z + bias), GELU'(z + bias)) in one pass."""
    _need(z, "z", torch.float32)
    _need(bias, "bias", torch.float32, ndim=1, last=z.shape[-1])
    h, d = torch.empty_like(z), torch.empty_like(z)
    _call(z.device, "upp_bias_gelu_fwd_d", _abi.ptr(z), _abi.ptr(bias), _abi.ptr(h), _abi.ptr(d), z.numel() // z.shape[-1], z.shape[-1])
    return h, d


def bias_gelu_bwd(g_h, z, bias):
    g_z = torch.empty_like(z)
    _call(z.device, "upp_bias_gelu_bwd", _abi.ptr(g_h), _abi.ptr(z), _abi.ptr(bias), _abi.ptr(g_z), z.numel() // z.shape[-1], z.shape[-1])
    return g_z


def ln_param_grad(g_h, xo, mean, rstd, chunks=32):
    rows = g_h.numel() // g_h.shape[-1]
    D = g_h.shape[-1]
    part = torch.empty((2, chunks, D), dtype=torch.float32, device=g_h.device)
    _call(g_h.device, "upp_ln_param_grad", _abi.ptr(g_h), _abi.ptr(xo), _abi.ptr(mean), _abi.ptr(rstd), _abi.ptr(part), rows, D, chunks)
    return part          # (2, chunks, D): [0] d_gamma partials, [1] d_beta partials; the caller sums over the chunks


def attn_fwd(qkv, B, L, H, scale):
    _need(qkv, "qkv", torch.float32)
    hd = qkv.numel() // (B * L * 3 * H)
    ctx = torch.empty((B, L, H * hd), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    _call(qkv.device, "upp_attn_fwd", _abi.ptr(qkv), _abi.ptr(ctx), _abi.ptr(lse), B, L, H, hd, float(scale))
    return ctx, lse


def attn_bwd(qkv, ctx, d_ctx, lse, B, L, H, scale):
    hd = qkv.numel() // (B * L * 3 * H)
    d_qkv = torch.empty_like(qkv)
    _call(qkv.device, "upp_attn_bwd", _abi.ptr(qkv), _abi.ptr(ctx), _abi.ptr(d_ctx), _abi.ptr(lse), _abi.ptr(d_qkv), B, L, H, hd, float(scale))
    return d_qkv


# ------------------------------------------------------------------ prompt propagation
def prop_pool_fwd(X, i1, u, keep, groups):
    D = X.shape[-1]
    pooled = torch.empty((groups, D), dtype=torch.float32, device=X.device)
    amax = torch.empty((groups, D), dtype=torch.uint8, device=X.device)
    _call(X.device, "upp_prop_pool_fwd", _abi.ptr(X), _abi.ptr(i1), _abi.ptr(u), float(keep), _abi.ptr(pooled), _abi.ptr(amax), groups, D)
    return pooled, amax


def prop_pool_bwd(g_pooled, amax, i1, u, keep, rows):
    groups, D = g_pooled.shape
    g_X = torch.empty((rows, D), dtype=torch.float32, device=g_pooled.device)
    _call(g_pooled.device, "upp_prop_pool_bwd", _abi.ptr(g_pooled), _abi.ptr(amax), _abi.ptr(i1), _abi.ptr(u), float(keep),
          _abi.ptr(g_X), rows, groups, D)
    return g_X


def prop_interp_fwd(X, lc, i2, idx8, w8, B, Lp, T, G2):
    D = X.shape[-1]
    out = torch.empty_like(X)
    _call(X.device, "upp_prop_interp_fwd", _abi.ptr(X), _abi.ptr(lc), _abi.ptr(i2), _abi.ptr(idx8), _abi.ptr(w8), _abi.ptr(out), B, Lp, T, G2, D)
    return out


def prop_interp_bwd(g_out, i2, idx8, w8, B, Lp, T, G2):
    D = g_out.shape[-1]
    g_c2 = torch.empty((B * G2, D), dtype=torch.float32, device=g_out.device)
    g_X = torch.empty_like(g_out)
    _call(g_out.device, "upp_prop_interp_bwd", _abi.ptr(g_out), _abi.ptr(i2), _abi.ptr(idx8), _abi.ptr(w8), _abi.ptr(g_c2), _abi.ptr(g_X),
          B, Lp, T, G2, D)
    return g_c2, g_X


def csr_build(keys, rows, seg_len=None, seg_rows=0):
    """Stable inverse of an int32 index list -> (start (rows+1), perm (n)); see upp_csr_build."""
    _need(keys, "keys", torch.int32)
    n = keys.numel()
    start = torch.empty(rows + 1, dtype=torch.int32, device=keys.device)
    perm = torch.empty(n, dtype=torch.int32, device=keys.device)
    _call(keys.device, "upp_csr_build", _abi.ptr(keys), n, int(seg_len or n), int(seg_rows), int(rows), _abi.ptr(start), _abi.ptr(perm))
    return start, perm


def prop_index(c1, c2, i1, i2, gather_idx, Lp, off, eps):
    """-> (i1a, i2a int32 absolute rows, idx8 int32 (B,T,8), w8 f32 (B,T,8)); see upp_prop_index."""
    _need(c1, "center1", torch.float32, 3, 3)
    _need(c2, "center2", torch.float32, 3, 3)
    _need(i1, "center1_idx", torch.int64)
    _need(i2, "center2_idx", torch.int64)
    B, T, _ = c1.shape
    G2 = c2.shape[1]
    if i1.numel() != B * G2 * 8 or i2.numel() != B * G2:
        raise RuntimeError("prop_index: index tensors do not match the centre shapes")
    dev = c1.device
    i1a = torch.empty(B * G2 * 8, dtype=torch.int32, device=dev)
    i2a = torch.empty(B * G2, dtype=torch.int32, device=dev)
    idx8 = torch.empty((B, T, 8), dtype=torch.int32, device=dev)
    w8 = torch.empty((B, T, 8), dtype=torch.float32, device=dev)
    _call(dev, "upp_prop_index", _abi.ptr(c1), _abi.ptr(c2), _abi.ptr(i1), _abi.ptr(i2), int(bool(gather_idx)), B, T, G2, int(Lp), int(off),
          float(eps), _abi.ptr(i1a), _abi.ptr(i2a), _abi.ptr(idx8), _abi.ptr(w8))
    return i1a, i2a, idx8, w8


def prop_fwd(X, i1, u, keep, i2, idx8, w8, gamma, beta, running_mean, running_var, momentum, eps, training, B, Lp, T, G2):
    D = X.shape[-1]
    groups = B * G2
    dev = X.device
    pooled = torch.empty((groups, D), dtype=torch.float32, device=dev)
    amax = torch.empty((groups, D), dtype=torch.uint8, device=dev)
    part = torch.empty(_abi.load().upp_prop_part_floats(groups, D), dtype=torch.float32, device=dev)
    mean = torch.empty(D, dtype=torch.float32, device=dev)
    rstd = torch.empty(D, dtype=torch.float32, device=dev)
    out = torch.empty_like(X)
    _call(dev, "upp_prop_fwd", _abi.ptr(X), _abi.ptr(i1), _abi.ptr(u), float(keep), _abi.ptr(i2), _abi.ptr(idx8), _abi.ptr(w8),
          _abi.ptr(gamma), _abi.ptr(beta), _abi.ptr(running_mean), _abi.ptr(running_var), float(momentum), float(eps), int(bool(training)),
          _abi.ptr(pooled), _abi.ptr(amax), _abi.ptr(part), _abi.ptr(mean), _abi.ptr(rstd), _abi.ptr(out), B, Lp, T, G2, D)
    return out, pooled, amax, mean, rstd


def prop_bwd(g_out, pooled, amax, mean, rstd, gamma, u, keep, w8, csr1, csr2, csr8, training, B, Lp, T, G2):
    D = g_out.shape[-1]
    groups = B * G2
    dev = g_out.device
    g_c2 = torch.empty((groups, D), dtype=torch.float32, device=dev)
    part = torch.empty(_abi.load().upp_prop_part_floats(groups, D), dtype=torch.float32, device=dev)
    g_gamma = torch.empty(D, dtype=torch.float32, device=dev)
    g_beta = torch.empty(D, dtype=torch.float32, device=dev)
    g_X = torch.empty_like(g_out)
    _call(dev, "upp_prop_bwd", _abi.ptr(g_out), _abi.ptr(pooled), _abi.ptr(amax), _abi.ptr(mean), _abi.ptr(rstd), _abi.ptr(gamma), _abi.ptr(u),
          float(keep), _abi.ptr(w8), _abi.ptr(csr1[0]), _abi.ptr(csr1[1]), _abi.ptr(csr2[0]), _abi.ptr(csr2[1]), _abi.ptr(csr8[0]),
          _abi.ptr(csr8[1]), int(bool(training)), _abi.ptr(g_c2), _abi.ptr(part), _abi.ptr(g_gamma), _abi.ptr(g_beta), _abi.ptr(g_X),
          B, Lp, T, G2, D)
    return g_X, g_gamma, g_beta


def prop_w8_grad(g_out, X, pooled, bn_stats, i2, idx8, B, Lp, T, G2):
    """Gradient w.r.t. the interpolation weights (B,T,8); bn_stats = (mean, rstd, gamma, beta) saved by prop_fwd, or None when `pooled`
    already holds the post-BatchNorm rows (prop_interp).  See upp_prop_w8_grad."""
    _need(g_out, "g_out", torch.float32)
    _need(X, "X", torch.float32)
    _need(pooled, "pooled", torch.float32)
    D = g_out.shape[-1]
    if g_out.numel() != B * Lp * D or X.numel() != B * Lp * D or pooled.numel() != B * G2 * D or idx8.numel() != B * T * 8 or i2.numel() != B * G2:
        raise RuntimeError("prop_w8_grad: operand shapes do not match (B, L', T, G2, D)")
    g_w8 = torch.empty((B, T, 8), dtype=torch.float32, device=g_out.device)
    st = bn_stats if bn_stats is not None else (None, None, None, None)
    _call(g_out.device, "upp_prop_w8_grad", _abi.ptr(g_out), _abi.ptr(X), _abi.ptr(pooled), _abi.ptr(st[0]), _abi.ptr(st[1]), _abi.ptr(st[2]),
          _abi.ptr(st[3]), _abi.ptr(i2), _abi.ptr(idx8), _abi.ptr(g_w8), B, Lp, T, G2, D)
    return g_w8


def prop_weights_bwd(c1, c2, idx8, g_w8, eps):
    """(g_c1 (B,T,3), g_c2 (B,G2,3)) from the gradient of the interpolation weights; see upp_prop_weights_bwd."""
    _need(c1, "c1", torch.float32, ndim=3)
    _need(c2, "c2", torch.float32, ndim=3)
    _need(g_w8, "g_w8", torch.float32)
    B, T, _ = c1.shape
    G2 = c2.shape[1]
    if c1.shape[2] != 3 or c2.shape[2] != 3 or c2.shape[0] != B or idx8.numel() != B * T * 8 or g_w8.numel() != B * T * 8 or idx8.dtype != torch.int32:
        raise RuntimeError("prop_weights_bwd: operand shapes do not match")
    g_c1, g_c2 = torch.empty_like(c1), torch.empty_like(c2)
    _call(c1.device, "upp_prop_weights_bwd", _abi.ptr(c1), _abi.ptr(c2), _abi.ptr(idx8), _abi.ptr(g_w8), float(eps), _abi.ptr(g_c1), _abi.ptr(g_c2),
          B, T, G2)
    return g_c1, g_c2


# ------------------------------------------------------------------ row operators of the frozen prompter branches
def _drop_seed(drop):
    """drop = (p, seed int64 device scalar, seed_add, salt) -> checked tuple for upp_bn_rows_drop_fwd / _bwd."""
    p, seed, add, salt = drop
    if not (isinstance(seed, torch.Tensor) and seed.is_cuda and seed.dtype == torch.int64 and seed.numel() == 1):
        raise RuntimeError("dropout seed must be a one-element int64 HIP (cuda) tensor (the layer's num_batches_tracked)")
    return float(p), seed, int(add), int(salt) & 0xFFFFFFFF


def bn_rows_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, training, relu, want_stats=False, drop=None):
    """drop = (p, seed, seed_add, salt): nn.Dropout(p) on the output in the same pass (training only; upp_bn_rows_drop_fwd)."""
    _need(x, "x", torch.float32, ndim=2)
    R, C = x.shape
    dev = x.device
    part = torch.empty(_abi.load().upp_bn_rows_part_floats(R, C), dtype=torch.float32, device=dev) if training else None
    mean = torch.empty(C, dtype=torch.float32, device=dev)
    rstd = torch.empty(C, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    if drop is not None and drop[0] > 0.0:
        if not training:
            raise RuntimeError("bn_rows_fwd: dropout belongs to the training-mode form")
        p, seed, add, salt = _drop_seed(drop)
        _call(dev, "upp_bn_rows_drop_fwd", _abi.ptr(x), _abi.ptr(gamma), _abi.ptr(beta), _abi.ptr(running_mean), _abi.ptr(running_var),
              float(momentum), float(eps), int(bool(relu)), p, _abi.ptr(seed), add, salt, _abi.ptr(part), _abi.ptr(mean), _abi.ptr(rstd), _abi.ptr(y), R, C)
    else:
        _call(dev, "upp_bn_rows_fwd", _abi.ptr(x), _abi.ptr(gamma), _abi.ptr(beta), _abi.ptr(running_mean), _abi.ptr(running_var),
              float(momentum), float(eps), int(bool(training)), int(bool(relu)), _abi.ptr(part), _abi.ptr(mean), _abi.ptr(rstd), _abi.ptr(y), R, C)
    return (y, mean, rstd) if want_stats else y


def bn_rows_bwd(x, g, mean, rstd, gamma, beta, relu, want_gx=True, drop=None):
    """Backward of the training-mode bn_rows_fwd: -> (g_x or None, g_gamma, g_beta).  drop: (p, seed, seed_add, salt) -- the forward's mask."""
    _need(x, "x", torch.float32, ndim=2)
    _need(g, "g", torch.float32, ndim=2)
    R, C = x.shape
    dev = x.device
    _need(mean, "mean", torch.float32, ndim=1, last=C)
    _need(rstd, "rstd", torch.float32, ndim=1, last=C)
    if g.shape != x.shape or any(t is not None and (t.shape != (C,) or t.dtype != torch.float32 or not t.is_cuda) for t in (gamma, beta)):
        raise RuntimeError("bn_rows_bwd: g must match x, gamma / beta must be (C,) f32 HIP tensors")
    part = torch.empty(_abi.load().upp_bn_rows_part_floats(R, C), dtype=torch.float32, device=dev)
    g_gamma = torch.empty(C, dtype=torch.float32, device=dev)
    g_beta = torch.empty(C, dtype=torch.float32, device=dev)
    g_x = torch.empty_like(x) if want_gx else None
    if drop is not None and drop[0] > 0.0:
        p, seed, add, salt = _drop_seed(drop)
        _call(dev, "upp_bn_rows_drop_bwd", _abi.ptr(x), _abi.ptr(g), _abi.ptr(mean), _abi.ptr(rstd), _abi.ptr(gamma), _abi.ptr(beta), int(bool(relu)),
              p, _abi.ptr(seed), add, salt, _abi.ptr(part), _abi.ptr(g_gamma), _abi.ptr(g_beta), _abi.ptr(g_x), R, C)
    else:
        _call(dev, "upp_bn_rows_bwd", _abi.ptr(x), _abi.ptr(g), _abi.ptr(mean), _abi.ptr(rstd), _abi.ptr(gamma), _abi.ptr(beta), int(bool(relu)),
              _abi.ptr(part), _abi.ptr(g_gamma), _abi.ptr(g_beta), _abi.ptr(g_x), R, C)
    return g_x, g_gamma, g_beta


def sqdist_topk(q, src, k):
    """q (B,N,3), src (B,S,3) -> (dist (B,N,k) f32, idx (B,N,k) int64): k nearest by the reference's square_distance form."""
    _need(q, "q", torch.float32, ndim=3, last=3)
    _need(src, "src", torch.float32, ndim=3, last=3)
    _same_device(q, src)
    B, N, _ = q.shape
    S = src.shape[1]
    if src.shape[0] != B:
        raise RuntimeError("q / src batch sizes differ")
    dist = torch.empty((B, N, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((B, N, k), dtype=torch.int64, device=q.device)
    _call(q.device, "upp_sqdist_topk", _abi.ptr(q), _abi.ptr(src), _abi.ptr(dist), _abi.ptr(idx), B, N, S, int(k))
    return dist, idx


def interp_fwd(dist, idx, feat, k, eps, out=None, col0=0):
    """dist / idx: (B,N,S') views whose last dim is contiguous (a sorted neighbour table); feat (B,S,C)."""
    _need(feat, "feat", torch.float32, ndim=3)
    B, S, C = feat.shape
    N = dist.shape[1]
    for t, name, dt in ((dist, "dist", torch.float32), (idx, "idx", torch.int64)):
        if not t.is_cuda or t.dtype != dt or t.dim() != 3 or t.stride(2) != 1 or t.stride(0) != N * t.stride(1):
            raise RuntimeError(f"{name} must be a HIP {dt} (B,N,S') table with contiguous rows")
    if dist.stride(1) != idx.stride(1) or dist.shape[:2] != idx.shape[:2] or dist.shape[0] != B:
        raise RuntimeError("dist / idx must share shape and row stride")
    if out is None:
        out = torch.empty((B, N, C), dtype=torch.float32, device=feat.device)
    _need(out, "out", torch.float32, ndim=3)
    _call(feat.device, "upp_interp_fwd", _abi.ptr(dist), _abi.ptr(idx), dist.stride(1), _abi.ptr(feat), _abi.ptr(out), out.shape[-1], int(col0),
          B, N, S, C, int(k), float(eps))
    return out


def interp_affine_fwd(dist, idx, feat, x3, wt, k, eps):
    """interp_fwd into a dense (B,N,C) matrix plus x3 (B,N,3) . wt (3,C); k <= 4, C >= 256, C % 4 == 0."""
    _need(feat, "feat", torch.float32, ndim=3)
    _need(x3, "x3", torch.float32, ndim=3, last=3)
    B, S, C = feat.shape
    _need(wt, "wt", torch.float32, ndim=2, last=C)
    N = dist.shape[1]
    for t, name, dt in ((dist, "dist", torch.float32), (idx, "idx", torch.int64)):
        if not t.is_cuda or t.dtype != dt or t.dim() != 3 or t.stride(2) != 1 or t.stride(0) != N * t.stride(1):
            raise RuntimeError(f"{name} must be a HIP {dt} (B,N,S') table with contiguous rows")
    if dist.stride(1) != idx.stride(1) or dist.shape[:2] != idx.shape[:2] or dist.shape[0] != B or x3.shape[:2] != dist.shape[:2] or wt.shape[0] != 3:
        raise RuntimeError("dist / idx / x3 / wt shapes do not match")
    out = torch.empty((B, N, C), dtype=torch.float32, device=feat.device)
    _call(feat.device, "upp_interp_affine_fwd", _abi.ptr(dist), _abi.ptr(idx), dist.stride(1), _abi.ptr(feat), _abi.ptr(x3), _abi.ptr(wt),
          _abi.ptr(out), B, N, S, C, int(k), float(eps))
    return out


def interp_bwd(dist, idx, g_out, S, k, eps):
    """Gradient of interp_fwd w.r.t. feat: g_out (B,N,C) -> (B,S,C); dist / idx as in interp_fwd."""
    _need(g_out, "g_out", torch.float32, ndim=3)
    B, N, C = g_out.shape
    for t, name, dt in ((dist, "dist", torch.float32), (idx, "idx", torch.int64)):
        if not t.is_cuda or t.dtype != dt or t.dim() != 3 or t.stride(2) != 1 or t.stride(0) != N * t.stride(1):
            raise RuntimeError(f"{name} must be a HIP {dt} (B,N,S') table with contiguous rows")
    if dist.stride(1) != idx.stride(1) or dist.shape[:2] != idx.shape[:2] or tuple(dist.shape[:2]) != (B, N):
        raise RuntimeError("dist / idx must share shape and row stride and match g_out")
    g_feat = torch.empty((B, S, C), dtype=torch.float32, device=g_out.device)
    _call(g_out.device, "upp_interp_bwd", _abi.ptr(dist), _abi.ptr(idx), dist.stride(1), _abi.ptr(g_out), C, 0, _abi.ptr(g_feat),
          B, N, int(S), C, int(k), float(eps))
    return g_feat


def interp_geo_bwd(dist, idx, feat, g_out, xyz1, xyz2, k, eps, need1=True, need2=True):
    """Gradient of interp_fwd w.r.t. the geometry behind the neighbour table: -> (g_xyz1 (B,N,3) or None, g_xyz2 (B,S,3) or None);
    upp_interp_geo_bwd."""
    _need(g_out, "g_out", torch.float32, ndim=3)
    _need(feat, "feat", torch.float32, ndim=3)
    _need(xyz1, "xyz1", torch.float32, ndim=3, last=3)
    _need(xyz2, "xyz2", torch.float32, ndim=3, last=3)
    B, N, C = g_out.shape
    S = feat.shape[1]
    for t, name, dt in ((dist, "dist", torch.float32), (idx, "idx", torch.int64)):
        if not t.is_cuda or t.dtype != dt or t.dim() != 3 or t.stride(2) != 1 or t.stride(0) != N * t.stride(1):
            raise RuntimeError(f"{name} must be a HIP {dt} (B,N,S') table with contiguous rows")
    if dist.stride(1) != idx.stride(1) or tuple(dist.shape[:2]) != (B, N) or tuple(xyz1.shape[:2]) != (B, N) or tuple(xyz2.shape[:2]) != (B, S) or feat.shape[2] != C:
        raise RuntimeError("interp_geo_bwd: shapes do not match")
    dev = g_out.device
    g1 = torch.empty((B, N, 3), dtype=torch.float32, device=dev) if need1 else None
    g2 = torch.empty((B, S, 3), dtype=torch.float32, device=dev) if need2 else None
    contrib = torch.empty((B * N, int(k), 3), dtype=torch.float32, device=dev) if need2 else None
    _call(dev, "upp_interp_geo_bwd", _abi.ptr(dist), _abi.ptr(idx), dist.stride(1), _abi.ptr(feat), _abi.ptr(g_out), C, 0, _abi.ptr(xyz1), _abi.ptr(xyz2),
          B, N, S, C, int(k), float(eps), _abi.ptr(g1), _abi.ptr(g2), _abi.ptr(contrib))
    return g1, g2


def posenc_fwd(x, freqs, out=None, col0=0):
    _need(x, "x", torch.float32, last=3)
    F = len(freqs)
    rows = x.numel() // 3
    width = 3 * (2 * F + 1)
    if out is None:
        out = torch.empty(x.shape[:-1] + (width,), dtype=torch.float32, device=x.device)
    _need(out, "out", torch.float32)
    import ctypes
    arr = (ctypes.c_float * max(F, 1))(*[float(f) for f in freqs])
    _call(x.device, "upp_posenc_fwd", _abi.ptr(x), arr, F, _abi.ptr(out), out.shape[-1], int(col0), rows)
    return out


# ------------------------------------------------------------------ classification tail
def cls_pool_fwd(x, gamma, beta, eps):
    _need(x, "x", torch.float32, ndim=3)
    B, L, D = x.shape
    dev = x.device
    feat = torch.empty((B, 2 * D), dtype=torch.float32, device=dev)
    amax = torch.empty((B, D), dtype=torch.int32, device=dev)
    mean = torch.empty(B * L, dtype=torch.float32, device=dev)
    rstd = torch.empty(B * L, dtype=torch.float32, device=dev)
    _call(dev, "upp_cls_pool_fwd", _abi.ptr(x), _abi.ptr(gamma), _abi.ptr(beta), float(eps), _abi.ptr(feat), _abi.ptr(amax), _abi.ptr(mean),
          _abi.ptr(rstd), B, L, D)
    return feat, amax, mean, rstd


def cls_pool_bwd(g_feat, x, mean, rstd, gamma, amax):
    _need(g_feat, "g_feat", torch.float32, ndim=2)
    B, L, D = x.shape
    g_x = torch.empty_like(x)
    _call(x.device, "upp_cls_pool_bwd", _abi.ptr(g_feat), _abi.ptr(x), _abi.ptr(mean), _abi.ptr(rstd), _abi.ptr(gamma), _abi.ptr(amax),
          _abi.ptr(g_x), B, L, D)
    return g_x


def ce_acc(logits, labels):
    _need(logits, "logits", torch.float32, ndim=2)
    _need(labels, "labels", torch.int64, ndim=1)
    B, C = logits.shape
    out2 = torch.empty(2, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    _call(logits.device, "upp_ce_acc", _abi.ptr(logits), _abi.ptr(labels), _abi.ptr(out2), _abi.ptr(dlogits), B, C)
    return out2, dlogits


def bn_relu_drop_fwd(z, gamma, beta, running_mean, running_var, momentum, eps, training, u, p):
    _need(z, "z", torch.float32, ndim=2)
    R, C = z.shape
    a = torch.empty_like(z)
    mean = torch.empty(C, dtype=torch.float32, device=z.device)
    rstd = torch.empty(C, dtype=torch.float32, device=z.device)
    _call(z.device, "upp_bn_relu_drop_fwd", _abi.ptr(z), _abi.ptr(gamma), _abi.ptr(beta), _abi.ptr(running_mean), _abi.ptr(running_var),
          float(momentum), float(eps), int(bool(training)), _abi.ptr(u), float(p), _abi.ptr(a), _abi.ptr(mean), _abi.ptr(rstd), R, C)
    return a, mean, rstd


def bn_relu_drop_bwd(g_a, z, gamma, beta, mean, rstd, training, u, p):
    R, C = z.shape
    g_z = torch.empty_like(z)
    g_gamma = torch.empty(C, dtype=torch.float32, device=z.device)
    g_beta = torch.empty(C, dtype=torch.float32, device=z.device)
    _call(z.device, "upp_bn_relu_drop_bwd", _abi.ptr(g_a), _abi.ptr(z), _abi.ptr(gamma), _abi.ptr(beta), _abi.ptr(mean), _abi.ptr(rstd),
          int(bool(training)), _abi.ptr(u), float(p), _abi.ptr(g_z), _abi.ptr(g_gamma), _abi.ptr(g_beta), R, C)
    return g_z, g_gamma, g_beta


# ------------------------------------------------------------------ optimizer tail
def batched_sum(jobs):
    """jobs: list of (part 2-D f32, column offset, rows n, length, row stride, dst f32, accumulate):
    dst (+)= sum over the n rows of part[:, offset:offset+length], every job in one launch (upp_batched_sum).  dst: `length` contiguous
    elements, or a 2-D WINDOW (rows, w) with rows * w == length, unit column stride and any row pitch (a column range of a wider matrix)."""
    if not jobs:
        return
    import ctypes
    k = len(jobs)
    dw, dp = [], []
    for part, off, n, length, ld, dst, acc in jobs:
        if not (isinstance(dst, torch.Tensor) and dst.is_cuda and dst.dtype == torch.float32):
            raise RuntimeError("batched_sum: dst must be an f32 HIP (cuda) tensor; upp_hip has no CPU path")
        if not part.is_cuda or part.dtype != torch.float32 or part.dim() != 2 or part.stride(1) != 1 or part.stride(0) != ld:
            raise RuntimeError("batched_sum: part must be a HIP f32 matrix with contiguous rows")
        if n != part.shape[0] or off < 0 or off + length > part.shape[1] or dst.numel() != length:
            raise RuntimeError("batched_sum: job geometry does not match its tensors")
        if dst.is_contiguous():
            dw.append(0), dp.append(0)
        elif dst.dim() == 2 and dst.stride(1) == 1 and dst.stride(0) >= dst.shape[1]:
            dw.append(dst.shape[1]), dp.append(dst.stride(0))
        else:
            raise RuntimeError("batched_sum: dst must be contiguous or a 2-D window with contiguous rows")
    src = (ctypes.c_void_p * k)(*[j[0].data_ptr() + 4 * j[1] for j in jobs])
    dst = (ctypes.c_void_p * k)(*[j[5].data_ptr() for j in jobs])
    n = (ctypes.c_int * k)(*[j[2] for j in jobs])
    ln = (ctypes.c_int * k)(*[j[3] for j in jobs])
    ld = (ctypes.c_int * k)(*[j[4] for j in jobs])
    acc = (ctypes.c_int * k)(*[int(bool(j[6])) for j in jobs])
    windows = any(dw)
    _call(jobs[0][0].device, "upp_batched_sum", src, dst, n, ln, ld, acc, (ctypes.c_int * k)(*dw) if windows else None,
          (ctypes.c_int * k)(*dp) if windows else None, k)


def copy_batched(dsts, srcs):
    """dst_j.copy_(src_j) for contiguous same-shape / same-dtype HIP tensors, every pair in one launch (upp_copy_batched)."""
    if not dsts:
        return
    import ctypes
    k = len(dsts)
    if len(srcs) != k:
        raise RuntimeError("copy_batched: as many sources as destinations")
    for d, s_ in zip(dsts, srcs):
        if not (d.is_cuda and s_.is_cuda and d.dtype == s_.dtype and d.shape == s_.shape and d.is_contiguous() and s_.is_contiguous()):
            raise RuntimeError("copy_batched: contiguous HIP (cuda) tensors of equal shape and dtype are required")
    S = (ctypes.c_void_p * k)(*[t.data_ptr() for t in srcs])
    D = (ctypes.c_void_p * k)(*[t.data_ptr() for t in dsts])
    n = (ctypes.c_longlong * k)(*[t.numel() * t.element_size() for t in dsts])
    _call(dsts[0].device, "upp_copy_batched", S, D, n, k)


def transpose_batched(pairs):
    """pairs: list of (src (rows, cols) contiguous f32, dst (cols, rows) contiguous f32): dst = src^T for all of them in one launch
    (upp_transpose_batched_f32)."""
    if not pairs:
        return
    import ctypes
    k = len(pairs)
    for src, dst in pairs:
        _need(src, "src", torch.float32, ndim=2)
        _need(dst, "dst", torch.float32, ndim=2)
        if tuple(dst.shape) != (src.shape[1], src.shape[0]):
            raise RuntimeError("transpose_batched: dst must be (cols, rows) of src")
    srcs = (ctypes.c_void_p * k)(*[a.data_ptr() for a, _ in pairs])
    dsts = (ctypes.c_void_p * k)(*[b.data_ptr() for _, b in pairs])
    rows = (ctypes.c_int * k)(*[a.shape[0] for a, _ in pairs])
    cols = (ctypes.c_int * k)(*[a.shape[1] for a, _ in pairs])
    _call(pairs[0][0].device, "upp_transpose_batched_f32", srcs, dsts, rows, cols, k)


def adamw_flat(p, g, m, v, n, split, state, scratch, lr, beta1, beta2, eps, weight_decay, max_norm):
    for t, name in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (state, "state"), (scratch, "scratch")):
        _need(t, name, torch.float32)
    _call(p.device, "upp_adamw_flat", _abi.ptr(p), _abi.ptr(g), _abi.ptr(m), _abi.ptr(v), int(n), int(split), _abi.ptr(state),
          _abi.ptr(scratch), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), float(max_norm))


# ------------------------------------------------------------------ bottleneck adapter
def adapter_fwd(ha, x, W1, b1, W2, b2, u, p, scale):
    D = ha.shape[-1]
    R = ha.numel() // D
    H = W1.shape[0]
    out = torch.empty_like(x)
    s1 = torch.empty((R, H), dtype=torch.float32, device=ha.device)
    _call(ha.device, "upp_adapter_fwd", _abi.ptr(ha), _abi.ptr(x), _abi.ptr(W1), _abi.ptr(b1), _abi.ptr(W2), _abi.ptr(b2), _abi.ptr(u),
          float(p), float(scale), _abi.ptr(out), _abi.ptr(s1), R, D, H)
    return out, s1


def ln_adapter_fwd(x, y, ybias, u, keep, mode, P, gamma, beta, eps, W1, b1, W2, b2, ud, p, scale, Lout):
    """Residual + prompt strip + adapter LayerNorm + adapter in one launch -> (out, xo, mean, rstd, s1); upp_ln_adapter_fwd."""
    _need(x, "x", torch.float32, ndim=3)
    B, Lin, D = x.shape
    H = W1.shape[0]
    dev = x.device
    for t_, n_, shp in ((gamma, "gamma", (D,)), (beta, "beta", (D,)), (W1, "W1", (H, D)), (b1, "b1", (H,)), (W2, "W2", (D, H)), (b2, "b2", (D,))):
        _need(t_, n_, torch.float32)
        if tuple(t_.shape) != shp:
            raise RuntimeError(f"ln_adapter_fwd: {n_} must be {shp}, got {tuple(t_.shape)}")
    xo = torch.empty((B, Lout, D), dtype=torch.float32, device=dev)
    out = torch.empty((B, Lout, D), dtype=torch.float32, device=dev)
    mean = torch.empty((B, Lout), dtype=torch.float32, device=dev)
    rstd = torch.empty((B, Lout), dtype=torch.float32, device=dev)
    s1 = torch.empty((B * Lout, H), dtype=torch.float32, device=dev)
    _call(dev, "upp_ln_adapter_fwd", _abi.ptr(x), _abi.ptr(y), _abi.ptr(ybias), _abi.ptr(u), float(keep), int(mode), int(P), _abi.ptr(gamma),
          _abi.ptr(beta), float(eps), _abi.ptr(W1), _abi.ptr(b1), _abi.ptr(W2), _abi.ptr(b2), _abi.ptr(ud), float(p), float(scale),
          _abi.ptr(xo), _abi.ptr(mean), _abi.ptr(rstd), _abi.ptr(s1), _abi.ptr(out), B, Lin, Lout, D, H)
    return out, xo, mean, rstd, s1


def ln_adapter_bwd(g_out, xo, mean, rstd, gamma, beta, s1, W1, W2, u, p, scale):
    """upp_adapter_bwd with the LayerNorm output rebuilt from the saved rows and statistics -> (g_ha, partials)."""
    D = xo.shape[-1]
    R = xo.numel() // D
    H = W1.shape[0]
    g_ha = torch.empty_like(xo)
    n = int(_abi.load().upp_adapter_part_floats(R, D))
    nblk = (R + 31) // 32
    part = torch.empty((nblk, n // nblk), dtype=torch.float32, device=xo.device)
    _call(xo.device, "upp_ln_adapter_bwd", _abi.ptr(g_out), _abi.ptr(xo), _abi.ptr(mean), _abi.ptr(rstd), _abi.ptr(gamma), _abi.ptr(beta),
          _abi.ptr(s1), _abi.ptr(W1), _abi.ptr(W2), _abi.ptr(u), float(p), float(scale), _abi.ptr(g_ha), _abi.ptr(part), R, D, H)
    return g_ha, part


def ln_adapter_bwd_fused(g_out, xo, mean, rstd, gamma, beta, s1, W1, W2, ud, p, scale, u, keep, mode, P, Lin, need_x, need_y, need_adapter,
                         need_ln, factors=False):
    """Backward of ln_adapter_fwd in one launch -> (g_x, g_y, adapter partials, LayerNorm partials); upp_ln_adapter_bwd_fused.
    factors=True: the third result is `fac` (B*Lout, 2 H) = [ga | d] per row instead of the per-workgroup partial matrices
    (upp_ln_adapter_bwd_factors; the weight gradients are then formed by adapter_wgrad_batched)."""
    B, Lout, D = xo.shape
    H = W1.shape[0]
    dev = xo.device
    R = B * Lout
    nwg = (R + 15) // 16
    g_x = torch.empty((B, Lin, D), dtype=torch.float32, device=dev) if need_x else None
    g_y = torch.empty((B, Lin, D), dtype=torch.float32, device=dev) if need_y else None
    ln_part = torch.empty((nwg, 2 * D), dtype=torch.float32, device=dev) if need_ln else None
    if factors:
        part = torch.empty((R, 2 * H), dtype=torch.float32, device=dev)
    else:
        part = torch.empty((nwg, int(_abi.load().upp_ln_adapter_part_floats(R, D)) // nwg), dtype=torch.float32, device=dev) if need_adapter else None
    _call(dev, "upp_ln_adapter_bwd_factors" if factors else "upp_ln_adapter_bwd_fused", _abi.ptr(g_out), _abi.ptr(xo), _abi.ptr(mean), _abi.ptr(rstd),
          _abi.ptr(gamma), _abi.ptr(beta), _abi.ptr(s1), _abi.ptr(W1), _abi.ptr(W2), _abi.ptr(ud), float(p), float(scale), _abi.ptr(u), float(keep),
          int(mode), int(P), _abi.ptr(g_x), _abi.ptr(g_y), _abi.ptr(part), _abi.ptr(ln_part), B, Lin, Lout, D, H)
    return g_x, g_y, part, ln_part


def adapter_wgrad_batched(jobs, H=32):
    """jobs: list of (xo (R, D), mean (R), rstd (R), gamma, beta, g_out (R, D), fac (R, 2 H), scale) -- the saved rows and statistics of a
    block's tail, the gradient that reached it and the factors ln_adapter_bwd_fused(factors=True) wrote.  ONE launch (per 16 jobs) forms
    every block's [dW1 | dW2 | db1 | db2] over `splits` row ranges -> list of (splits, 2 H D + H + D) partial matrices, to be summed over
    their rows (batched_sum); upp_adapter_wgrad_batched."""
    if not jobs:
        return []
    import ctypes
    k = len(jobs)
    lib = _abi.load()
    D = jobs[0][0].shape[-1]
    for xo, mean, rstd, gamma, beta, g_out, fac, _ in jobs:
        R = xo.numel() // D
        for t, name, n in ((xo, "xo", R * D), (g_out, "g_out", R * D), (mean, "mean", R), (rstd, "rstd", R), (gamma, "gamma", D), (beta, "beta", D),
                           (fac, "fac", R * 2 * H)):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
                raise RuntimeError("adapter_wgrad_batched: %s must be a contiguous f32 HIP tensor of %d elements" % (name, n))
    rows = [j[0].numel() // D for j in jobs]
    splits = int(lib.upp_adapter_wgrad_splits(max(rows)))
    psz = 2 * H * D + H + D
    parts = [torch.empty((splits, psz), dtype=torch.float32, device=jobs[0][0].device) for _ in jobs]
    arr = lambda i: (ctypes.c_void_p * k)(*[j[i].data_ptr() for j in jobs])
    _call(jobs[0][0].device, "upp_adapter_wgrad_batched", arr(0), arr(1), arr(2), arr(3), arr(4), arr(5), arr(6), (ctypes.c_int * k)(*rows),
          (ctypes.c_float * k)(*[float(j[7]) for j in jobs]), (ctypes.c_void_p * k)(*[p_.data_ptr() for p_ in parts]), k, splits, D, H)
    return parts


def adapter_bwd(g_out, ha, s1, W1, W2, u, p, scale):
    D = ha.shape[-1]
    R = ha.numel() // D
    H = W1.shape[0]
    g_ha = torch.empty_like(ha)
    n = int(_abi.load().upp_adapter_part_floats(R, D))
    nblk = (R + 31) // 32
    part = torch.empty((nblk, n // nblk), dtype=torch.float32, device=ha.device)
    _call(ha.device, "upp_adapter_bwd", _abi.ptr(g_out), _abi.ptr(ha), _abi.ptr(s1), _abi.ptr(W1), _abi.ptr(W2), _abi.ptr(u),
          float(p), float(scale), _abi.ptr(g_ha), _abi.ptr(part), R, D, H)
    return g_ha, part
