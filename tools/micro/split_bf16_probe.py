"""EXPERIMENT: f32 Linear on the bf16 matrix pipe (operands split exactly into three bf16 terms, six MFMA products).

Builds tools/micro/split_bf16_gemm.hip into tools/micro/build/libsplit_bf16.so, then prints, per Linear shape of
the training step: time of the library f32 GEMM, of the LDS-staged split kernel (tiles 11..22) and of the LDS-free
fragment-order kernel (f11..f24); errors against f64 (max |C - C64| / max |C64|, and the componentwise error in units
of 2^-24 |A|.|W|^T); and the "ceiling" probes of the fragment kernel (operands loaded once / no split arithmetic).

    python tools/micro/split_bf16_probe.py          # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "iccv2025-upp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from upp_hip.build import FLAGS as PRODUCT_FLAGS  # noqa: E402  (the flag list of the shipped library, -packed-fp32-ops included: a micro build measures the same code)
import torch.nn.functional as F  # noqa: E402

from bench import time_kernel  # noqa: E402


def build():
    out = os.path.join(HERE, "build", "libsplit_bf16.so")
    src = os.path.join(HERE, "split_bf16_gemm.hip")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + PRODUCT_FLAGS + [
                               "-shared", src, "-o", out])
    lib = ctypes.CDLL(out)
    vp, i = ctypes.c_void_p, ctypes.c_int
    lib.upp_split_bf16.argtypes = [vp, i, i, i, i, vp, vp]
    lib.upp_split_frag_elems.restype = ctypes.c_longlong
    lib.upp_split_frag_elems.argtypes = [i, i]
    lib.upp_split_bf16_frag.argtypes = [vp, i, i, i, i, vp, vp]
    for name in ("upp_linear_split", "upp_linear_frag"):
        getattr(lib, name).argtypes = [vp, i, vp, vp, vp, i, vp, i, i, i, i, i, vp]
    return lib


LIB = None


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def split_planes(W):
    planes = torch.empty((3,) + tuple(W.shape), dtype=torch.bfloat16, device=W.device)
    assert LIB.upp_split_bf16(_p(W), W.shape[0], W.shape[1], W.shape[1], 0, _p(planes), _st()) == 0
    return planes


def split_frag(W, transpose=False):
    N, K = (W.shape[1], W.shape[0]) if transpose else tuple(W.shape)
    data = torch.empty(LIB.upp_split_frag_elems(N, K), dtype=torch.bfloat16, device=W.device)
    assert LIB.upp_split_bf16_frag(_p(W), N, K, W.shape[1], 1 if transpose else 0, _p(data), _st()) == 0
    return data, N, K


def linear(kind, x, w, N, bias=None, gelu=False):
    K = x.shape[-1]
    M = x.numel() // K
    out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    fn = LIB.upp_linear_split if kind == "split" else LIB.upp_linear_frag
    rc = fn(_p(x), K, _p(w), _p(bias), _p(out), N, None, N, M, N, K, 1 if gelu else 0, _st())
    assert rc == 0, rc
    return out


SHAPES = [(2400, 1152, 384), (2400, 384, 384), (2400, 1536, 384), (2400, 384, 1536), (65536, 512, 256), (65536, 384, 512)]


def main():
    global LIB
    LIB = build()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    print("%-18s %5s %8s %8s | %9s %9s | %8s %8s" % ("M,N,K", "tile", "us", "TFLOP/s", "rel", "rel f32", "ulp", "ulp f32"))
    for M, N, K in SHAPES:
        A = torch.randn(M, K, device=dev, generator=g) * torch.exp(torch.randn(M, 1, device=dev, generator=g))
        W = torch.randn(N, K, device=dev, generator=g) * 0.05
        b = torch.randn(N, device=dev, generator=g)
        planes = split_planes(W)
        assert (planes.double().sum(0) - W.double()).abs().max().item() == 0.0      # hi + mid + lo == W exactly
        frag, _, _ = split_frag(W)
        C64 = A.double() @ W.double().t() + b.double()
        bound = A.double().abs() @ W.double().abs().t() + b.double().abs()

        def err(C):
            d = (C.double() - C64).abs()
            return (d.max() / C64.abs().max()).item(), (d / bound).max().item() * 2 ** 24
        flops = 2.0 * M * N * K
        t = time_kernel(lambda: F.linear(A, W, b))
        r, u = err(F.linear(A, W, b))
        print("%-18s %5s %8.1f %8.1f | %9s %9.2e | %8s %8.2f" % ("%d,%d,%d" % (M, N, K), "lib", t * 1e3, flops / t / 1e9, "", r, "", u))
        for kind, w, setter, tiles in (("split", planes, LIB.upp_gemm_split_set_tile, (11, 12, 21, 22)),
                                       ("frag", frag, LIB.upp_gemm_frag_set_tile, (11, 12, 21, 22, 24))):
            for tile in tiles:
                setter(tile)
                r, u = err(linear(kind, A, w, N, b))
                t = time_kernel(lambda: linear(kind, A, w, N, b))
                print("%-18s %5s %8.1f %8.1f | %9.2e %9s | %8.2f" % ("", ("f%d" if kind == "frag" else "%d") % tile, t * 1e3,
                                                                     flops / t / 1e9, r, "", u))
            setter(0)
        mfma_us = 6.0 * M * N * K / 16384 * 32 / 1024 / 2.4e3
        for tile in (11, 12, 22):
            row = []
            for mode in (0, 100, 200, 300):
                LIB.upp_gemm_frag_set_tile(mode + tile)
                row.append(time_kernel(lambda: linear("frag", A, frag, N)) * 1e3)
            print("   ceiling f%d (MFMA issue floor %.1f us): full %.1f | operands loaded once %.1f | no split arithmetic %.1f | neither %.1f us"
                  % (tile, mfma_us, *row))
        LIB.upp_gemm_frag_set_tile(0)
    # input-gradient operand (transposed planes) and the GELU epilogue
    M, N, K = 2400, 1536, 384
    A = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    ft, n2, k2 = split_frag(W, transpose=True)
    gy = torch.randn(M, N, device=dev, generator=g)
    gx = linear("frag", gy, ft, n2)
    ref = gy.double() @ W.double()
    print("dgrad (transposed planes): max err %.2e, f32 library %.2e" % ((gx.double() - ref).abs().max().item(),
                                                                       ((gy @ W).double() - ref).abs().max().item()))
    frag, _, _ = split_frag(W)
    h = linear("frag", A, frag, N, b, gelu=True)
    print("bias+GELU epilogue: max err %.2e" % (h.double() - F.gelu(F.linear(A.double(), W.double(), b.double()))).abs().max().item())


if __name__ == "__main__":
    main()
