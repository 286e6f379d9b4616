"""upp_linear_f32 (csrc/linear.hip): the Transformer blocks' Linear layers (reference models/Point_MAE_pretask_dev.py:153-196)
on the FP32 matrix cores, against plain fp32 torch (tolerance 1e-5 rel of the output scale: north_star) and an f64 product."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from upp_hip import functional as HF, ops, _abi

pytestmark = pytest.mark.gpu

TOKENS = [2400, 2080, 2048, 1120]                                                   # B*L of the classification step (B = 32)
LAYERS = [("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536)]   # (name, out, in)
CONFIGS = [0x4412, 0x4311, 0x3411, 0x2421, 0x2321, 0x2241, 0x1241]


def close(a, b, rtol=1e-5, atol_scale=2e-6):
    a, b = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_scale * max(np.abs(b).max(), 1e-30))


def _operands(M, N, K, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) * K ** -0.5
    b = torch.randn(N, device='cuda', generator=g)
    return a, w, b


@pytest.mark.parametrize("M", TOKENS)
@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_forward_and_data_gradient_at_every_block_shape(M, layer):
    _, N, K = layer
    a, w, b = _operands(M, N, K, seed=M + N)
    a.requires_grad_(True)
    out = HF.linear(a, w, b)
    gy = torch.randn_like(out)
    out.backward(gy)
    g_ours = a.grad.clone()
    a.grad = None
    ref = F.linear(a, w, b)
    ref.backward(gy)
    close(out, ref)
    close(g_ours, a.grad)
    # and against the exact product: the k-ordered f32 fma chains are as accurate as the library's
    exact = a.detach().double() @ w.double().t() + b.double()
    assert (out.double() - exact).abs().max() <= 2.0 * (ref.double() - exact).abs().max() + 1e-6 * exact.abs().max()


@pytest.mark.parametrize("cfg", CONFIGS, ids=[hex(c) for c in CONFIGS])
@pytest.mark.parametrize("shape", [(75, 384, 384), (2400, 1152, 384), (333, 96, 256), (1, 40, 128), (129, 1536, 512)])
def test_every_compiled_decomposition_gives_the_same_product(cfg, shape):
    M, N, K = shape
    if K % (32 * ((cfg >> 4) & 15) * (cfg & 15)):
        pytest.skip("k-stage does not divide K")
    a, w, _ = _operands(M, N, K, seed=cfg)
    close(ops.linear_f32(a, w, tile=cfg), F.linear(a, w))


def test_epilogues_bias_gelu_and_saved_derivative():
    M, N, K = 2400, 1536, 384
    a, w, b = _operands(M, N, K, seed=3)
    z = (a.double() @ w.double().t() + b.double())
    close(ops.linear_f32(a, w, b, ops.LIN_BIAS), z.float())
    close(ops.linear_f32(a, w, b, ops.LIN_BIAS_GELU), F.gelu(z).float(), atol_scale=1e-6)
    h, d = ops.linear_f32(a, w, b, ops.LIN_BIAS_GELU_D)
    close(h, F.gelu(z).float(), atol_scale=1e-6)
    zz = z.clone().requires_grad_(True)
    F.gelu(zz).sum().backward()
    close(d, zz.grad.float(), atol_scale=3e-6)          # |GELU'| <= 1.13: 3.4e-6 absolute incl. the f32 rounding of z itself
    z32 = ops.linear_f32(a, w, b, ops.LIN_BIAS).double().requires_grad_(True)      # the epilogue alone, from the kernel's own z
    F.gelu(z32).sum().backward()
    close(d, z32.grad.float(), atol_scale=5e-7)
    close(h, F.gelu(z32.detach()).float(), atol_scale=5e-7)
    fac = torch.randn(M, N, device='cuda')
    close(ops.linear_f32(a, w, None, ops.LIN_MUL, aux=fac), ((a.double() @ w.double().t()) * fac.double()).float())


@pytest.mark.parametrize("M", [2400, 1120])
def test_frozen_mlp_forward_backward_equals_the_torch_composition(M):
    g = torch.Generator(device='cuda').manual_seed(M)
    x = torch.randn(M, 384, device='cuda', generator=g, requires_grad=True)
    w1 = torch.randn(1536, 384, device='cuda', generator=g) * 0.05
    b1 = torch.randn(1536, device='cuda', generator=g) * 0.1
    w2 = torch.randn(384, 1536, device='cuda', generator=g) * 0.03
    gm = torch.randn(M, 384, device='cuda', generator=g)
    m = HF.mlp_gelu(x, w1, b1, w2)
    m.backward(gm)
    g_ours = x.grad.clone()
    x.grad = None
    ref = F.linear(F.gelu(F.linear(x, w1, b1)), w2)
    ref.backward(gm)
    close(m, ref)
    close(g_ours, x.grad)
    with torch.no_grad():
        close(HF.mlp_gelu(x, w1, b1, w2), ref)


def test_transposed_weight_cache_follows_in_place_updates():
    a, w, _ = _operands(64, 384, 384, seed=9)
    a.requires_grad_(True)
    HF.linear(a, w).sum().backward()
    first = a.grad.clone()
    a.grad = None
    with torch.no_grad():
        w.mul_(2.0)                                  # e.g. load_state_dict: same storage, new version
    HF.linear(a, w).sum().backward()
    close(a.grad, 2.0 * first)


def test_trainable_weight_gets_all_three_gradients():
    a, w, b = _operands(300, 256, 384, seed=11)
    for t in (a, w, b):
        t.requires_grad_(True)
    gy = torch.randn(300, 256, device='cuda')
    HF.linear(a, w, b).backward(gy)
    got = [t.grad.clone() for t in (a, w, b)]
    for t in (a, w, b):
        t.grad = None
    F.linear(a, w, b).backward(gy)
    for x, t in zip(got, (a, w, b)):
        close(x, t.grad, atol_scale=5e-6)


def test_strided_rows_and_rejections():
    lib = _abi.load()
    base = torch.randn(100, 3, 384, device='cuda')
    a = base[:, 1]                                     # rows 3*384 floats apart: served without a copy
    w = torch.randn(96, 384, device='cuda') * 0.05
    close(ops.linear_f32(a, w), F.linear(a, w))
    with pytest.raises(RuntimeError):
        ops.linear_f32(torch.randn(8, 100, device='cuda'), torch.randn(16, 100, device='cuda'))     # K % 32 != 0
    with pytest.raises(RuntimeError):
        ops.linear_f32(torch.randn(8, 64), torch.randn(16, 64))                                     # CPU tensors: no CPU path
    assert lib.upp_linear_f32(None, 0, None, 0, None, None, 0, None, 0, 1, 1, 32, 0, 0, None) == -1
    assert lib.upp_linear_f32(a.data_ptr(), 384, w.data_ptr(), 384, None, a.data_ptr(), 96, None, 0, 4, 96, 384, 0, 0x9999, None) == -2
