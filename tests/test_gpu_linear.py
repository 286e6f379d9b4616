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
RT_TALL = 0x2512211        # csrc/linear_rt.hip: 2 x 2 waves of 2 x 2 blocks, 2 LDS stages of 32 (the low byte reads KS = 1, KC = 1)
CONFIGS = [0x4412, 0x4311, 0x3411, 0x2421, 0x2321, 0x2241, 0x1241, 0x2221, RT_TALL]


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
        ops.linear_f32(torch.randn(8, 102, device='cuda'), torch.randn(16, 102, device='cuda'))     # K % 4 != 0
    with pytest.raises(RuntimeError):
        ops.linear_f32(torch.randn(8, 64), torch.randn(16, 64))                                     # CPU tensors: no CPU path
    assert lib.upp_linear_f32(None, 0, None, 0, None, None, 0, None, 0, 1, 1, 32, 0, 0, None) == -1
    assert lib.upp_linear_f32(a.data_ptr(), 384, w.data_ptr(), 384, None, a.data_ptr(), 96, None, 0, 4, 96, 384, 0, 0x9999, None) == -2


@pytest.fixture
def exact_wgrad():
    """The bit-pinned weight-gradient kernel (linear_rt.hip); the default since round 4 is the split-bf16 one (tests/test_gpu_linear_sb.py)."""
    saved = ops.WGRAD_SPLIT_BF16
    ops.WGRAD_SPLIT_BF16 = False
    yield
    ops.WGRAD_SPLIT_BF16 = saved


@pytest.mark.parametrize("shape", [(2400, 1152, 384), (2400, 384, 1536), (65536, 384, 512), (75, 40, 96), (1, 4, 4), (33, 256, 128),
                                   (4160, 1536, 384), (1000, 132, 36)])
def test_weight_gradient_partials_sum_to_the_product(shape, exact_wgrad):
    """upp_linear_wgrad_f32: dW = G^T . X, rows split over workgroups, partial tiles summed in split order."""
    import oracle as O
    M, N, K = shape
    g = torch.Generator(device='cuda').manual_seed(M + N)
    G = torch.randn(M, N, device='cuda', generator=g)
    X = torch.randn(M, K, device='cuda', generator=g)
    part = ops.linear_wgrad(G, X)
    assert part.shape[1:] == (N, K) and part.shape[0] == _abi.load().upp_linear_wgrad_splits(M, N, K)
    if M * N * K <= 2400 * 1152 * 384:          # bit for bit against the CPU restatement of the kernel's order (one ascending-row chain per run)
        rows = -(-M // part.shape[0])
        rows = -(-rows // 32) * 32
        np.testing.assert_array_equal(part.cpu().numpy(), O.linear_wgrad(G.cpu().numpy(), X.cpu().numpy(), rows))
    ref = G.double().t() @ X.double()
    got = part.double().sum(0)
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=2e-6 * ref.abs().max().item())
    lib = (G.t() @ X).double()
    assert (got - ref).abs().max() <= 2.0 * (lib - ref).abs().max() + 1e-6 * ref.abs().max()
    # strided operands (column windows of wider matrices) are served in place
    wide = torch.randn(M, N + 8, device='cuda', generator=g)
    part2 = ops.linear_wgrad(wide[:, 4:4 + N], X)
    close(part2.sum(0), wide[:, 4:4 + N].t() @ X, atol_scale=5e-6)


def test_grouped_weight_gradients_equal_the_single_launches_and_the_oracle(exact_wgrad):
    """upp_linear_wgrad_grouped_f32: the weight gradients of a backward pass in one launch -- mixed shapes, ragged edges, strided
    operands; every partial bit for bit the CPU restatement (one ascending-row fmaf chain per run of rows)."""
    import oracle as O
    g = torch.Generator(device='cuda').manual_seed(77)
    shapes = [(2080, 1152, 384), (864, 384, 1536), (2080, 384, 384), (1000, 132, 36), (77, 52, 36), (4100, 256, 128), (33, 4, 4)]
    pairs = []
    for M, N, K in shapes:
        wide = torch.randn(M, N + 8, device='cuda', generator=g)
        pairs.append((wide[:, 4:4 + N], torch.randn(M, K, device='cuda', generator=g)))
    parts = ops.linear_wgrad_grouped(pairs)
    for (G, X), part, (M, N, K) in zip(pairs, parts, shapes):
        rows = -(-(-(-M // part.shape[0])) // 32) * 32
        np.testing.assert_array_equal(part.cpu().numpy(), O.linear_wgrad(G.cpu().numpy(), X.cpu().numpy(), rows))
        close(part.sum(0), G.t() @ X, atol_scale=5e-6)
    # inside a deferred scope: queued, launched once at exit, accumulated into the registered gradient buffers
    ws = [torch.zeros(N, K, device='cuda', requires_grad=True) for _, N, K in shapes[:3]]
    bufs = {w.data_ptr(): torch.ones(w.numel(), device='cuda') for w in ws}
    with HF.deferred_sums(bufs) as scope:
        for w, (G, X) in zip(ws, pairs[:3]):
            assert HF.weight_grad(G, X, w) is None
    assert scope.routed == set(bufs)
    for w, (G, X) in zip(ws, pairs[:3]):
        close(bufs[w.data_ptr()].view_as(w), 1.0 + G.t() @ X, atol_scale=5e-6)


def _reference_encoder(enc, pg):
    """The reference's Encoder.forward (models/Point_MAE_unify.py:204-222) on the module's own Conv1d / BatchNorm1d layers."""
    bs, g, n, _ = pg.shape
    x = pg.reshape(bs * g, n, 3).transpose(2, 1)
    f = enc.first_conv(x)
    fg = torch.max(f, dim=2, keepdim=True)[0]
    f = torch.cat([fg.expand(-1, -1, n), f], dim=1)
    f = enc.second_conv(f)
    return torch.max(f, dim=2, keepdim=False)[0].reshape(bs, g, enc.encoder_channel)


@pytest.mark.parametrize("trainable", [True, False])
def test_patch_embedding_with_a_gradient_runs_on_our_kernels_and_matches_torch_autograd(trainable):
    """Pre-training trains the patch embedding (all parameter gradients); stage 2 of the UPP recipe differentiates THROUGH the
    frozen one (gradient of the input points, BatchNorm in training mode).  Forward, data, weight, bias and BatchNorm
    gradients against torch autograd on the reference formulation -- and no library GEMM in between."""
    from models.upp_layers import Encoder
    torch.manual_seed(3)
    enc = Encoder(384).cuda().train()
    for p in enc.parameters():
        p.requires_grad_(trainable)
    pg = (0.3 * torch.randn(4, 64, 32, 3, device='cuda')).requires_grad_(True)
    gy = torch.randn(4, 64, 384, device='cuda')
    stats0 = {k: v.clone() for k, v in enc.state_dict().items() if 'running' in k}

    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        out = enc(pg)
        out.backward(gy)
    torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert any('linear_f32_kernel' in k for k in names) and (not trainable or any('wgrad_grouped_kernel' in k or 'wgrad_sb_kernel' in k for k in names))
    assert not any(k.startswith('Cijk') for k in names), [k for k in names if k.startswith('Cijk')]
    got = {'x': pg.grad.clone()}
    got.update({n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None})
    stats1 = {k: v.clone() for k, v in enc.state_dict().items() if 'running' in k}
    pg.grad = None
    for p in enc.parameters():
        p.grad = None
    enc.load_state_dict({**enc.state_dict(), **stats0})
    ref = _reference_encoder(enc, pg)
    ref.backward(gy)
    close(out, ref, atol_scale=5e-6)
    # The two max-pools route a gradient to ONE point per (group, channel): where the two largest pre-pool values of a column differ by
    # less than the f32 rounding of a formulation, the arg-max -- and with it a few entries of the input gradient and their share of the
    # row sums behind every parameter gradient -- legitimately flips (~0.3 % of the entries, in BOTH f32 formulations: the reference
    # concatenates [global | local] for one 512-wide GEMM, this path splits it).  Arbitrated in f64: every gradient of this path is as
    # close to an f64 evaluation of the reference formulation as torch's f32 evaluation of it is.
    enc64 = Encoder(384).cuda().double().train()
    enc64.load_state_dict({k: v.double() for k, v in {**enc.state_dict(), **stats0}.items()})
    for p in enc64.parameters():
        p.requires_grad_(trainable)
    pg64 = pg.detach().double().requires_grad_(True)
    _reference_encoder(enc64, pg64).backward(gy.double())
    exact = {'x': pg64.grad}
    exact.update({n: p.grad for n, p in enc64.named_parameters() if p.grad is not None})
    torch32 = {'x': pg.grad}
    torch32.update({n: p.grad for n, p in enc.named_parameters() if p.grad is not None})
    for n, r in exact.items():
        if trainable and n in ('first_conv.0.bias', 'first_conv.3.bias', 'second_conv.0.bias'):
            # a bias in front of a training-mode BatchNorm (first_conv.3.bias: through the linear 512 -> 512 layer) has gradient
            # exactly 0: both sides hold rounding noise only
            wmax = got[n.replace('bias', 'weight')].abs().max().item()
            assert got[n].abs().max().item() < 1e-3 * wmax and torch32[n].abs().max().item() < 1e-3 * wmax
            continue
        scale = r.abs().max().item()
        ea, eb = (torch32[n].double() - r).abs(), (got[n].double() - r).abs()
        assert eb.max().item() <= 4.0 * ea.max().item() + 1e-5 * scale, (n, eb.max().item(), ea.max().item(), scale)
        assert eb.mean().item() <= 3.0 * ea.mean().item() + 1e-6 * scale, (n, eb.mean().item(), ea.mean().item(), scale)
    assert trainable or set(got) == {'x'}
    for k in stats1:
        close(stats1[k], enc.state_dict()[k], atol_scale=1e-5)


@pytest.mark.parametrize("cfg", CONFIGS, ids=[hex(c) for c in CONFIGS])
def test_bit_exact_against_the_cpu_restatement_of_its_summation_order(cfg):
    """upp_linear_f32 == oracle.linear_f32 BIT FOR BIT: the FP32 MFMA is a k-ordered fmaf chain (lower lane half first), the
    kernel feeds k in a fixed permutation inside each group of 32, wave groups own fixed sub-chunks of every stage and their
    partial tiles are summed in group order.  Pins the arithmetic of every compiled decomposition, edge tiles included."""
    import oracle as O
    ks, kc = (cfg >> 4) & 15, cfg & 15
    for M, N, K in ((75, 96, 128 * max(1, ks * kc // 2)), (200, 40, 384), (333, 160, 1536 if (32 * ks * kc) <= 512 else 384)):
        if K % (32 * ks * kc):
            continue
        g = torch.Generator(device='cuda').manual_seed(cfg + M)
        a = torch.randn(M, K, device='cuda', generator=g)
        w = torch.randn(N, K, device='cuda', generator=g) * K ** -0.5
        b = torch.randn(N, device='cuda', generator=g)
        x = torch.randn(M, N, device='cuda', generator=g)
        an, wn = a.cpu().numpy(), w.cpu().numpy()
        np.testing.assert_array_equal(ops.linear_f32(a, w, tile=cfg).cpu().numpy(), O.linear_f32(an, wn, ks=ks, kc=kc))
        np.testing.assert_array_equal(ops.linear_f32(a, w, b, ops.LIN_BIAS, tile=cfg).cpu().numpy(),
                                      O.linear_f32(an, wn, bias=b.cpu().numpy(), ks=ks, kc=kc, epilogue=1))
        np.testing.assert_array_equal(ops.linear_f32(a, w, None, ops.LIN_MUL, aux=x, tile=cfg).cpu().numpy(),
                                      O.linear_f32(an, wn, aux=x.cpu().numpy(), ks=ks, kc=kc, epilogue=4))


@pytest.mark.parametrize("M,N,K", [(4100, 200, 96), (300, 50, 64), (65536, 256, 128), (129, 132, 36)])
def test_register_tiled_kernel_every_epilogue(M, N, K):
    """The tall-matrix kernel (csrc/linear_rt.hip) forced on small and ragged shapes: every epilogue against torch, the plain ones bit for
    bit against the oracle (ks = 1), rows that are not 16-byte aligned (N = 50: dword stores), and the library's own choice at 65,536 rows."""
    import oracle as O
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) * K ** -0.5
    b = torch.randn(N, device='cuda', generator=g)
    x = torch.randn(M, N, device='cuda', generator=g)
    if M >= 65536:
        assert _abi.load().upp_linear_tile(M, N, K) == RT_TALL
    if M * N * K <= 4100 * 200 * 96:
        an, wn = a.cpu().numpy(), w.cpu().numpy()
        np.testing.assert_array_equal(ops.linear_f32(a, w, tile=RT_TALL).cpu().numpy(), O.linear_f32(an, wn))
        np.testing.assert_array_equal(ops.linear_f32(a, w, b, ops.LIN_BIAS, tile=RT_TALL).cpu().numpy(), O.linear_f32(an, wn, bias=b.cpu().numpy(), epilogue=1))
        np.testing.assert_array_equal(ops.linear_f32(a, w, None, ops.LIN_MUL, aux=x, tile=RT_TALL).cpu().numpy(), O.linear_f32(an, wn, aux=x.cpu().numpy(), epilogue=4))
    ref = F.linear(a, w, b)
    close(ops.linear_f32(a, w, b, ops.LIN_BIAS, tile=RT_TALL), ref)
    close(ops.linear_f32(a, w, b, ops.LIN_BIAS_RELU, tile=RT_TALL), torch.relu(ref))
    close(ops.linear_f32(a, w, b, ops.LIN_BIAS_GELU, tile=RT_TALL), F.gelu(ref), atol_scale=5e-7 * 4)
    h, d = ops.linear_f32(a, w, b, ops.LIN_BIAS_GELU_D, tile=RT_TALL)
    z = ref.detach().clone().requires_grad_(True)
    F.gelu(z).sum().backward()
    close(h, F.gelu(ref), atol_scale=5e-7 * 4)
    close(d, z.grad, atol_scale=5e-7 * 4)
    close(ops.linear_f32(a, w, None, ops.LIN_MUL, aux=x, tile=RT_TALL), F.linear(a, w) * x)


@pytest.mark.parametrize("M,N,K", [(32, 40, 256), (32, 256, 40), (1024, 64, 12), (2400, 384, 100), (35072, 32, 60), (75, 96, 4), (1024, 16, 192)])
def test_contraction_lengths_that_are_not_a_multiple_of_the_k_stage(M, N, K):
    """K % 4 == 0 is enough: the last k-stage takes its granules beyond K from a page of zeros (bit-exact against the oracle on
    zero-padded operands, for the library's choice and for every compiled decomposition; strided operands included)."""
    import oracle as O
    lib = _abi.load()
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    big = torch.randn(M, K + 8, device='cuda', generator=g)
    a = big[:, 4:4 + K]                                       # row stride K + 8, 16-byte aligned start
    w = torch.randn(N, K, device='cuda', generator=g) * K ** -0.5
    b = torch.randn(N, device='cuda', generator=g)
    an, wn = a.cpu().numpy(), w.cpu().numpy()
    close(ops.linear_f32(a, w, b, ops.LIN_BIAS), F.linear(a, w, b))
    for cfg in [lib.upp_linear_tile(M, N, K)] + CONFIGS:
        ks, kc = (cfg >> 4) & 15, cfg & 15
        np.testing.assert_array_equal(ops.linear_f32(a, w, b, ops.LIN_BIAS, tile=cfg).cpu().numpy(),
                                      O.linear_f32(an, wn, bias=b.cpu().numpy(), ks=ks, kc=kc, epilogue=1))
    # and through autograd (data gradient: contraction over N)
    x = a.clone().requires_grad_(True)
    HF.linear(x, w, b).backward(torch.ones(M, N, device='cuda'))
    close(x.grad, torch.ones(M, N, device='cuda') @ w)


@pytest.mark.parametrize("M,N,K,act", [(1024, 128, 3, 'gelu'), (2048, 128, 3, 'gelu'), (35072, 32, 59, None), (35072, 32, 59, 'relu'), (1024, 64, 12, 'relu'),
                                        (7, 256, 64, None), (33, 5, 1, 'gelu'), (100, 200, 17, None)])
def test_small_k_linear(M, N, K, act):
    """upp_linear_smallk_f32 (K <= 64, N <= 256, any alignment) with bias and activation in the same pass; HF.linear routes there."""
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    x = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) * K ** -0.5
    b = torch.randn(N, device='cuda', generator=g)
    ref = F.linear(x, w, b)
    ref = F.relu(ref) if act == 'relu' else (F.gelu(ref) if act == 'gelu' else ref)
    close(ops.linear_smallk(x, w, b, {None: 0, 'relu': 1, 'gelu': 2}[act]), ref)
    with torch.no_grad():
        close(HF.linear(x, w, b, act=act), ref)
    if act != 'gelu':       # summation order: ascending k, one fmaf chain per output -- bit-exact against the CPU restatement
        import oracle as O
        np.testing.assert_array_equal(ops.linear_smallk(x, w, b, 1 if act == 'relu' else 0).cpu().numpy(),
                                      O.linear_smallk(x.cpu().numpy(), w.cpu().numpy(), b.cpu().numpy(), 1 if act == 'relu' else 0))


@pytest.mark.parametrize("act", [None, 'relu', 'gelu'])
def test_activation_epilogues_through_hf_linear(act):
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(1024, 384, device='cuda', generator=g)
    w = torch.randn(192, 384, device='cuda', generator=g) * 0.05
    b = torch.randn(192, device='cuda', generator=g)
    ref = F.linear(x, w, b)
    ref = F.relu(ref) if act == 'relu' else (F.gelu(ref) if act == 'gelu' else ref)
    with torch.no_grad():
        close(HF.linear(x, w, b, act=act), ref)
        close(HF.linear(x, w, None, act=act), _a(F.linear(x, w), act))
    xg = x.clone().requires_grad_(True)                      # with a gradient: the activation is torch's, the GEMMs ours
    wg = w.clone().requires_grad_(True)
    out = HF.linear(xg, wg, b, act=act, own_wgrad=True)
    close(out, ref)
    out.backward(torch.ones_like(out))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    _a(F.linear(xr, wr, b), act).backward(torch.ones_like(out))
    close(xg.grad, xr.grad, rtol=2e-5, atol_scale=1e-5)
    close(wg.grad, wr.grad, rtol=2e-5, atol_scale=1e-5)


def _a(y, act):
    return F.relu(y) if act == 'relu' else (F.gelu(y) if act == 'gelu' else y)


@pytest.mark.parametrize("rows,cols,ld", [(256, 768, 768), (40, 256, 256), (33, 65, 80), (1, 1, 1), (512, 256, 3456)])
def test_transpose_kernel(rows, cols, ld):
    big = torch.randn(rows, ld, device='cuda')
    w = big[:, :cols]
    assert torch.equal(ops.transpose(w), w.t().contiguous())


def test_block_shapes_bit_exact_against_the_oracle():
    """The library's own choice of decomposition at the four Linear layers of a block (B = 32, L = 75)."""
    import oracle as O
    lib = _abi.load()
    for _, N, K in LAYERS:
        M = 2400
        c = lib.upp_linear_tile(M, N, K)
        a, w, _ = _operands(M, N, K, seed=N + K)
        np.testing.assert_array_equal(ops.linear_f32(a, w).cpu().numpy(),
                                      O.linear_f32(a.cpu().numpy(), w.cpu().numpy(), ks=(c >> 4) & 15, kc=c & 15))


@pytest.mark.parametrize("rows,cols,off,length", [(65536, 512, 0, 512), (10000, 100, 4, 50), (4097, 64, 0, 64), (300, 40, 8, 16)])
def test_tall_column_sums_in_two_stages(rows, cols, off, length):
    """upp_colsum_partials (first stage of a bias gradient over very many rows) + the row sum of its partials == the column sum;
    and the deferred-sum front door takes that route by itself above functional._DeferredSums.TALL rows."""
    g = torch.Generator(device='cuda').manual_seed(rows + cols)
    part = torch.randn(rows, cols, device='cuda', generator=g)
    ref = part[:, off:off + length].double().sum(0)
    p = ops.colsum_partials(part, off, length)
    assert p.shape[1] == length and p.shape[0] >= 1
    torch.testing.assert_close(p.double().sum(0), ref, rtol=1e-5, atol=1e-4 * max(1.0, float(ref.abs().max())) * 1e-1)
    routed, t = HF._DEFERRED.reduce(0, part, off, length)          # no scope open: summed at once and returned
    assert not routed
    torch.testing.assert_close(t.double(), ref, rtol=1e-5, atol=1e-4 * max(1.0, float(ref.abs().max())) * 1e-1)


@pytest.mark.parametrize("M,N,K", [(34432, 32, 27), (34432, 64, 3), (5000, 32, 59), (33, 16, 7), (70000, 40, 12)])
def test_small_k_linear_with_gradients(M, N, K):
    """_LinearSmallK: forward / data gradient on upp_linear_smallk_f32, weight gradient on upp_linear_smallk_wgrad_f32 (partials summed by
    the caller), bias gradient a column sum -- against torch autograd."""
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    x = torch.randn(M, K, device='cuda', generator=g).requires_grad_(True)
    w = (torch.randn(N, K, device='cuda', generator=g) * K ** -0.5).requires_grad_(True)
    b = torch.randn(N, device='cuda', generator=g).requires_grad_(True)
    go = torch.randn(M, N, device='cuda', generator=g)
    out = HF.linear(x, w, b)
    gx, gw, gb = torch.autograd.grad(out, [x, w, b], go)
    xr, wr, br = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    ref = F.linear(xr, wr, br)
    rx, rw, rb = torch.autograd.grad(ref, [xr, wr, br], go)
    close(out, ref)
    close(gx, rx)
    close(gw, rw, rtol=2e-5, atol_scale=2e-5)
    close(gb, rb, rtol=2e-5, atol_scale=2e-5)
    part = ops.linear_smallk_wgrad(go, x.detach())
    close(part.sum(0), rw, rtol=2e-5, atol_scale=2e-5)


@pytest.mark.parametrize("M,N,K", [(2048, 128, 3), (1024, 128, 3), (35072, 3, 64), (1000, 3, 32)])
def test_narrow_layers_with_a_gradient_on_the_input_stay_on_own_kernels(M, N, K, monkeypatch, capsys):
    """Stage 2 of the recipe differentiates through frozen narrow layers: the 3 -> 128 first layer of the position MLPs (input = centres
    that carry a gradient: forward on the small-K kernel, data gradient on the matrix cores with W^T zero-padded to 4 output columns) and
    the 64 -> 3 score head (data gradient contracts over its 3 outputs: small-K kernel).  Against torch autograd; nothing declined."""
    monkeypatch.setenv("UPP_VERBOSE", "1")
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    x = torch.randn(M, K, device='cuda', generator=g).requires_grad_(True)
    w = torch.randn(N, K, device='cuda', generator=g) * K ** -0.5              # frozen
    b = torch.randn(N, device='cuda', generator=g)
    go = torch.randn(M, N, device='cuda', generator=g)
    HF._declined.clear()
    out = HF.linear(x, w, b)
    gx, = torch.autograd.grad(out, [x], go)
    xr = x.detach().clone().requires_grad_(True)
    ref = F.linear(xr, w, b)
    rx, = torch.autograd.grad(ref, [xr], go)
    close(out, ref)
    close(gx, rx)
    assert not HF._declined and "fused path declined" not in capsys.readouterr().err


@pytest.mark.parametrize("M,N,K,rows", [(65536, 512, 256, 32), (65536, 512, 1024, 2048), (32768, 512, 256, 64), (65536 + 96, 512, 256, 32)])
def test_group_bias_in_the_epilogue_of_the_tall_kernel(M, N, K, rows):
    """upp_linear_group_bias_f32: a bias per group of 2^s rows added in the register-tiled kernel's epilogue -- bit-identical to the plain
    product followed by the broadcast add (one f32 add per element either way), gradients against torch autograd."""
    g = torch.Generator(device='cuda').manual_seed(M + K)
    if M % rows:
        M -= M % rows
    x = torch.randn(M, K, device='cuda', generator=g).requires_grad_(True)
    w = (torch.randn(N, K, device='cuda', generator=g) * K ** -0.5).requires_grad_(True)
    gb = torch.randn(M // rows, N, device='cuda', generator=g).requires_grad_(True)
    assert ops.linear_group_bias_usable(M, N, K, rows)
    out = HF.linear_group_bias(x, w, gb, rows)
    assert type(out.grad_fn).__name__ == '_LinearGroupBiasBackward'
    plain = (ops.linear_f32(x.detach(), w.detach()).view(M // rows, rows, N) + gb.detach().unsqueeze(1)).view(M, N)
    assert torch.equal(out.detach(), plain)
    go = torch.randn(M, N, device='cuda', generator=g)
    gx, gw, ggb = torch.autograd.grad(out, [x, w, gb], go)
    xr, wr, br = (t.detach().clone().requires_grad_(True) for t in (x, w, gb))
    ref = (F.linear(xr, wr).view(M // rows, rows, N) + br.unsqueeze(1)).view(M, N)
    rx, rw, rb = torch.autograd.grad(ref, [xr, wr, br], go)
    close(gx, rx, rtol=2e-5, atol_scale=2e-5)
    close(gw, rw, rtol=2e-5, atol_scale=2e-5)
    close(ggb, rb, rtol=2e-5, atol_scale=2e-5)
    # not a tall problem / not a power of two: the broadcast add on the short kernels, same function
    xs = torch.randn(4096, K, device='cuda', generator=g)
    gs = torch.randn(4096 // 32, N, device='cuda', generator=g)
    assert not ops.linear_group_bias_usable(4096, N, K, 32)
    close(HF.linear_group_bias(xs, w.detach(), gs, 32), (F.linear(xs, w.detach()).view(-1, 32, N) + gs.unsqueeze(1)).view(4096, N))


def test_trainable_weight_transposes_follow_the_weights_inside_a_step_driver():
    """Inside a step driver (TRANSPOSED.managed) the W^T copy of a trainable weight is persistent and refreshed by ONE batched launch
    (upp_transpose_batched_f32) at the start of a step: the data gradient must follow in-place weight updates; outside a driver every
    use transposes afresh."""
    g = torch.Generator(device='cuda').manual_seed(9)
    ws = [(torch.randn(n, k, device='cuda', generator=g) * 0.05).requires_grad_(True) for n, k in ((96, 384), (384, 64), (40, 256))]
    xs = [torch.randn(50, w.shape[1], device='cuda', generator=g).requires_grad_(True) for w in ws]

    def grads():
        return [torch.autograd.grad(HF.linear(x, w).sum(), x)[0] for x, w in zip(xs, ws)]

    def refs():
        return [torch.ones(50, w.shape[0], device='cuda') @ w.detach() for w in ws]

    HF.TRANSPOSED.managed = True
    try:
        HF.TRANSPOSED.refresh_trainable()
        for a, b in zip(grads(), refs()):
            close(a, b)
        with torch.no_grad():
            for w in ws:
                w.mul_(1.5).add_(0.01)                       # what an optimizer step does, without telling anyone
        stale = grads()                                      # (persistent copies: still the old weights -- by design)
        assert not torch.allclose(stale[0], refs()[0], rtol=1e-3)
        HF.TRANSPOSED.refresh_trainable()                    # what TrainStep._forward_backward does first
        for a, b in zip(grads(), refs()):
            close(a, b)
    finally:
        HF.TRANSPOSED.managed = False
    with torch.no_grad():
        ws[0].mul_(0.5)
    close(grads()[0], refs()[0])                             # unmanaged: transposed at every use
    pairs = [(w.detach(), torch.empty(w.shape[1], w.shape[0], device='cuda')) for w in ws]
    ops.transpose_batched(pairs)
    for src, dst in pairs:
        assert torch.equal(dst, src.t().contiguous())
