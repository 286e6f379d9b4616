"""Furthest point sampling beside other work on the same GPU (the pipelined step's situation); parity with the oracle: test_gpu_parity.py."""
import pytest
import torch

import _seeded
from upp_hip import _abi, ops


@pytest.mark.gpu
def test_fps_is_bit_stable_beside_a_co_running_split_bf16_linear_stream():
    """Round 4: with packed-f32 VALU instructions in the round loop (v_pk_add_f32 ... op_sel:[0,1]) the picks changed in ~3 of 4 calls while a
    split-bf16 Linear workgroup shared the CU -- the instruction returned a - 0 in its low half (tools/micro/src/lds_canary.cpp) -- which
    is exactly the situation of the pipelined step (front-end FPS beside the back-end's GEMMs).  The library is built without packed-f32
    instructions (upp_hip/build.py); this is the regression test: every result beside the co-runner equals the result on an idle GPU."""
    s2 = torch.cuda.Stream()
    a = torch.randn(4096, 1536, device='cuda')
    w = torch.randn(384, 1536, device='cuda') * 1536 ** -0.5
    w._upp_persistent = True
    out = torch.empty(4096, 384, device='cuda')
    ops.linear_f32(a, w, out=out, frozen=True)
    assert ops.linear_sb_tile(4096, 384, 1536) > 0
    for N, M in ((1024, 1024), (1228, 1024), (1936, 1536)):
        x = _seeded.noisy_clouds(2, N, seed=5).cuda().contiguous()
        ref = ops.fps(x, M)
        torch.cuda.synchronize()
        for it in range(12):
            with torch.cuda.stream(s2):
                for _ in range(60):
                    ops.linear_f32(a, w, out=out, frozen=True)
            got = ops.fps(x, M)
            torch.cuda.synchronize()
            assert torch.equal(got, ref), (N, M, it)


@pytest.mark.gpu
def test_knn_and_chamfer_indices_are_bit_stable_beside_a_co_running_split_bf16_linear_stream():
    """The other exact-index operators of the path (kNN grouping, Chamfer arg-mins) had hipcc-packed f32 arithmetic as well before the
    library lost its packed-f32 instructions: their results beside the co-runner equal the results on an idle GPU, bit for bit."""
    s2 = torch.cuda.Stream()
    a = torch.randn(4096, 1536, device='cuda')
    w = torch.randn(384, 1536, device='cuda') * 1536 ** -0.5
    w._upp_persistent = True
    out = torch.empty(4096, 384, device='cuda')
    ops.linear_f32(a, w, out=out, frozen=True)
    x = _seeded.noisy_clouds(32, 1024, seed=9).cuda().contiguous()
    y = _seeded.noisy_clouds(32, 1024, seed=10).cuda().contiguous()
    _, cen = ops.fps(x, 64, want_centers=True)
    torch.cuda.synchronize()

    def run():
        d, ki, nb = ops.knn(x, cen, 32, want_dist=True, want_neigh=True)
        d1, d2, i1, i2 = ops.chamfer_fwd(x, y)
        return [t.clone() for t in (d, ki, nb, d1, d2, i1, i2)]
    ref = run()
    torch.cuda.synchronize()
    for it in range(10):
        with torch.cuda.stream(s2):
            for _ in range(40):
                ops.linear_f32(a, w, out=out, frozen=True)
        got = run()
        torch.cuda.synchronize()
        for g_, r_ in zip(got, ref):
            assert torch.equal(g_, r_), it


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(32, 1228, 1024), (32, 1024, 64), (32, 1024, 256), (8, 2048, 512), (6, 1552, 1536), (4, 1096, 64), (3, 1024, 64)])
@pytest.mark.parametrize("form", [(2, False), (2, True), (4, False), (4, True)])
def test_fps_packed_forms_return_the_indices_of_the_spread_form(shape, form):
    """ops.fps_form: 2 / 4 clouds per workgroup, with and without the whole-LDS reservation (the pipelined step's front-end uses (2, True)):
    bit-identical indices and centres; a batch the count does not divide (B = 6 with 4, B = 3) takes the next smaller form."""
    B, N, M = shape
    g = torch.Generator(device='cuda').manual_seed(N + M)
    x = torch.rand(B, N, 3, device='cuda', generator=g) * 2 - 1
    ref, cref = ops.fps(x, M, want_centers=True)
    with ops.fps_form(*form):
        got, cgot = ops.fps(x, M, want_centers=True)
    assert torch.equal(got, ref) and torch.equal(cgot, cref)
    lib = _abi.load()
    assert lib.upp_fps_ex(None, None, None, 1, 64, 8, 0x300, None) == -1 and lib.upp_fps_ex(None, None, None, 1, 64, 8, 0x2000, None) == -1
