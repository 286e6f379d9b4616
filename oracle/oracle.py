"""Python face of the CPU oracle (oracle/upp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  numpy in, numpy out.

Parity status: Chamfer / EMD restate the reference's in-tree CUDA and are pinned by the
reference's own known answers (tests/test_oracle_golden.py).  FPS / gather / kNN restate
third-party kernels that are not vendored in the reference (pointnet2_ops 3.0.0,
KNN_CUDA 0.2) and that the reference never tests: PARITY UNPINNED for those three.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libupp_oracle.so")
_lib = None

_f = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64 = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_int = ctypes.c_int


def build(force=False):
    src = os.path.join(_HERE, "upp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.oracle_fps_block_size.argtypes = [_int]
        L.oracle_fps.argtypes = [_f, _int, _int, _int, _i32, ctypes.c_void_p]
        L.oracle_gather.argtypes = [_f, _i32, _f, _int, _int, _int, _int]
        L.oracle_gather_grad.argtypes = [_f, _i32, _f, _int, _int, _int, _int]
        L.oracle_knn.argtypes = [_f, _f, _int, _int, _int, _int, ctypes.c_void_p, _i64]
        L.oracle_group.argtypes = [_f, _f, _i64, _f, _int, _int, _int, _int]
        L.oracle_chamfer_fwd.argtypes = [_f, _f, _int, _int, _int, _f, _f, _i32, _i32]
        L.oracle_chamfer_bwd.argtypes = [_f, _f, _i32, _i32, _f, _f, _int, _int, _int, _f, _f]
        L.oracle_emd_approxmatch.argtypes = [_f, _f, _int, _int, _int, _f]
        L.oracle_emd_matchcost.argtypes = [_f, _f, _f, _int, _int, _int, _f]
        L.oracle_emd_matchcost_grad.argtypes = [_f, _f, _f, _f, _int, _int, _int, _f, _f]
        L.oracle_linear_f32.argtypes = [_f, _f, ctypes.c_void_p, ctypes.c_void_p, _f, _int, _int, _int, _int, _int, _int]
        L.oracle_linear_smallk.argtypes = [_f, _f, ctypes.c_void_p, _f, _int, _int, _int, _int]
        L.oracle_linear_wgrad.argtypes = [_f, _f, _f, _int, _int, _int, _int]
        L.oracle_num_threads.restype = _int
        L.oracle_set_threads.argtypes = [_int]
        _lib = L
    return _lib


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


def num_threads():
    return lib().oracle_num_threads()


def set_threads(n):
    lib().oracle_set_threads(int(n))


def fps_block_size(n):
    return lib().oracle_fps_block_size(int(n))


def fps(xyz, npoint):
    xyz = _c(xyz)
    B, N, _ = xyz.shape
    idx = np.empty((B, npoint), dtype=np.int32)
    lib().oracle_fps(xyz, B, N, npoint, idx, None)
    return idx


def gather(feat, idx):
    feat, idx = _c(feat), _c(idx, np.int32)
    B, C, N = feat.shape
    M = idx.shape[1]
    out = np.empty((B, C, M), dtype=np.float32)
    lib().oracle_gather(feat, idx, out, B, C, N, M)
    return out


def gather_grad(grad_out, idx, N):
    grad_out, idx = _c(grad_out), _c(idx, np.int32)
    B, C, M = grad_out.shape
    g = np.empty((B, C, N), dtype=np.float32)
    lib().oracle_gather_grad(grad_out, idx, g, B, C, N, M)
    return g


def knn(ref, query, k, want_dist=True):
    ref, query = _c(ref), _c(query)
    B, N, _ = ref.shape
    Q = query.shape[1]
    idx = np.empty((B, Q, k), dtype=np.int64)
    dist = np.empty((B, Q, k), dtype=np.float32) if want_dist else None
    rc = lib().oracle_knn(ref, query, B, N, Q, k, dist.ctypes.data if want_dist else None, idx)
    if rc != 0:
        raise ValueError("oracle_knn: k must satisfy 1 <= k <= N")
    return dist, idx


def group(xyz, center, idx):
    xyz, center, idx = _c(xyz), _c(center), _c(idx, np.int64)
    B, N, _ = xyz.shape
    _, G, K = idx.shape
    out = np.empty((B, G, K, 3), dtype=np.float32)
    lib().oracle_group(xyz, center, idx, out, B, N, G, K)
    return out


def chamfer_fwd(xyz1, xyz2):
    xyz1, xyz2 = _c(xyz1), _c(xyz2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.empty((B, n), np.float32); d2 = np.empty((B, m), np.float32)
    i1 = np.empty((B, n), np.int32); i2 = np.empty((B, m), np.int32)
    lib().oracle_chamfer_fwd(xyz1, xyz2, B, n, m, d1, d2, i1, i2)
    return d1, d2, i1, i2


def chamfer_bwd(xyz1, xyz2, idx1, idx2, gd1, gd2):
    xyz1, xyz2 = _c(xyz1), _c(xyz2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.empty((B, n, 3), np.float32); g2 = np.empty((B, m, 3), np.float32)
    lib().oracle_chamfer_bwd(xyz1, xyz2, _c(idx1, np.int32), _c(idx2, np.int32), _c(gd1), _c(gd2), B, n, m, g1, g2)
    return g1, g2


def emd_approxmatch(xyz1, xyz2):
    xyz1, xyz2 = _c(xyz1), _c(xyz2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = np.empty((B, m, n), np.float32)
    lib().oracle_emd_approxmatch(xyz1, xyz2, B, n, m, match)
    return match


def emd_matchcost(xyz1, xyz2, match):
    xyz1, xyz2, match = _c(xyz1), _c(xyz2), _c(match)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = np.empty((B,), np.float32)
    lib().oracle_emd_matchcost(xyz1, xyz2, match, B, n, m, cost)
    return cost


def emd_matchcost_grad(grad_cost, xyz1, xyz2, match):
    xyz1, xyz2, match = _c(xyz1), _c(xyz2), _c(match)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.empty((B, n, 3), np.float32); g2 = np.empty((B, m, 3), np.float32)
    lib().oracle_emd_matchcost_grad(_c(grad_cost), xyz1, xyz2, match, B, n, m, g1, g2)
    return g1, g2


# ---------------------------------------------------------------- torch-facing stand-ins (tests only)
def linear_f32(a, w, bias=None, aux=None, ks=1, kc=1, epilogue=0):
    """C = a . w^T in the summation order of upp_linear_f32 for the decomposition (KS = ks wave groups, 32 ks kc values of k per
    stage); epilogue 0 none / 1 + bias / 4 * aux.  See oracle_linear_f32 in upp_oracle.c."""
    a, w = _c(a), _c(w)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and K % 4 == 0
    stage = 32 * ks * kc
    if K % stage:                  # the kernel's last k-stage reads zeros beyond K
        pad = stage - K % stage
        a, w = _c(np.pad(a, ((0, 0), (0, pad)))), _c(np.pad(w, ((0, 0), (0, pad))))
        K += pad
    out = np.empty((M, N), np.float32)
    b = _c(bias) if bias is not None else None
    x = _c(aux) if aux is not None else None
    lib().oracle_linear_f32(a, w, None if b is None else b.ctypes.data_as(ctypes.c_void_p),
                            None if x is None else x.ctypes.data_as(ctypes.c_void_p), out, M, N, K, ks, kc, epilogue)
    return out


def bf16_rne(x):
    """f32 -> the nearest bf16 (ties to even), returned as f32: v_cvt_pk_bf16_f32 (finite inputs)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3_bf16(x):
    """x (f32) -> (x1, x2, x3), each a bf16 value held in f32, with x1 + x2 + x3 == x exactly for x = 0 and |x| >= 2^-110 (absolute
    error below 2^-133 otherwise): x1 = rne(x), x2 = rne(x - x1), x3 = rne(x - x1 - x2) (both residuals are exact in f32).  The operand split of upp_linear_sb_f32 / upp_linear_sb_prep (csrc/linear_sb.hip)."""
    x = np.ascontiguousarray(x, np.float32)
    x1 = bf16_rne(x)
    r = x - x1
    x2 = bf16_rne(r)
    x3 = bf16_rne(r - x2)          # (exact, i.e. no rounding happens, for |x| >= 2^-110: below, x - x1 - x2 falls under the bf16 subnormal grid)
    return x1, x2, x3


def linear_split(a, w, bias=None):
    """C = a . w^T as upp_linear_sb_f32 forms it, in EXACT arithmetic (float64 sums of the exact bf16 x bf16 products): the six terms
    a1 w1 + a1 w2 + a2 w1 + a2 w2 + a1 w3 + a3 w1 of the three-way bf16 split of both operands.  What the kernel adds to this is only
    the rounding of its f32 accumulation (the summation order inside a bf16 MFMA is not documented, so the kernel is held to this by
    tolerance, not bit for bit).  Stands for torch.nn.functional.linear (reference models/Point_MAE_pretask_dev.py:153-196)."""
    a1, a2, a3 = (t.astype(np.float64) for t in split3_bf16(a))
    w1, w2, w3 = (t.astype(np.float64) for t in split3_bf16(w))
    c = a1 @ w1.T + (a1 @ w2.T + a2 @ w1.T) + (a2 @ w2.T + a1 @ w3.T + a3 @ w1.T)
    if bias is not None:
        c = c + np.asarray(bias, np.float64)
    return c


def linear_wgrad(g, x, rows):
    """(splits, N, K) partial weight gradients g (M,N)^T . x (M,K) as upp_linear_wgrad_grouped_f32 sums them: runs of `rows` rows,
    one ascending-row fmaf chain each.  See oracle_linear_wgrad in upp_oracle.c."""
    g, x = _c(g), _c(x)
    M, N = g.shape
    K = x.shape[1]
    assert x.shape[0] == M and rows > 0
    out = np.empty(((M + rows - 1) // rows, N, K), np.float32)
    lib().oracle_linear_wgrad(g, x, out, M, N, K, int(rows))
    return out


def linear_smallk(x, w, bias=None, act=0):
    """act(x . w^T + bias) as upp_linear_smallk_f32 sums it (ascending k, fmaf); act 0 none / 1 ReLU."""
    x, w = _c(x), _c(w)
    M, K = x.shape
    N = w.shape[0]
    out = np.empty((M, N), np.float32)
    b = _c(bias) if bias is not None else None
    lib().oracle_linear_smallk(x, w, None if b is None else b.ctypes.data_as(ctypes.c_void_p), out, M, N, K, int(act))
    return out


def torch_ops():
    """Oracle-backed callables with the signatures of models.upp_layers.OPS, for running the
    model on a GPU-less host in tests.  Differentiable w.r.t. the coordinates through torch
    gathers (indices come from the oracle)."""
    import torch

    def fps_gather(xyz, npoint):
        idx = torch.from_numpy(fps(xyz.detach().cpu().numpy(), int(npoint))).to(xyz.device)
        centers = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
        return centers, idx

    def knn_group(xyz, center, k):
        _, idx = knn(xyz.detach().cpu().numpy(), center.detach().cpu().numpy(), int(k), want_dist=False)
        idx = torch.from_numpy(idx).to(xyz.device)
        B, G, K = idx.shape
        nb = torch.gather(xyz, 1, idx.reshape(B, G * K, 1).expand(-1, -1, 3)).reshape(B, G, K, 3)
        return nb - center.unsqueeze(2), idx

    class _Chamfer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a, b):
            d1, d2, i1, i2 = chamfer_fwd(a.detach().numpy(), b.detach().numpy())
            ctx.save_for_backward(a, b, torch.from_numpy(i1), torch.from_numpy(i2))
            return torch.from_numpy(d1), torch.from_numpy(d2)

        @staticmethod
        def backward(ctx, g1, g2):
            a, b, i1, i2 = ctx.saved_tensors
            ga, gb = chamfer_bwd(a.detach().numpy(), b.detach().numpy(), i1.numpy(), i2.numpy(),
                                 g1.contiguous().numpy(), g2.contiguous().numpy())
            return torch.from_numpy(ga), torch.from_numpy(gb)

    return {"fps_gather": fps_gather, "knn_group": knn_group, "chamfer": _Chamfer.apply}
