"""upp_linear_sb_f32 / upp_linear_sb_prep (csrc/linear_sb.hip): the Linear layers of the Transformer blocks (reference
models/Point_MAE_pretask_dev.py:153-196) with frozen weights, f32 operands split exactly into three bf16 terms, six bf16 MFMA
products accumulated in f32.  Checked against float64 and against the exact-f32 kernel (upp_linear_f32) on the same operands; the
split arithmetic itself is restated in oracle.linear_split (tests/test_linear_split_oracle.py, no GPU)."""
import numpy as np
import pytest
import torch

import oracle
from upp_hip import functional as HF, ops, _abi

pytestmark = pytest.mark.gpu

def _compiled_tiles():
    """Every tile shape the library compiles: pick_sb's candidates (UPP_SB_CONFIGS, csrc/linear_sb.hip) and the tuned table's (csrc/linear_sb_tuned.h)."""
    import os
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "iccv2025-upp_amd", "upp_hip", "csrc")
    out = []
    for f, macro in (("linear_sb.hip", "UPP_SB_CONFIGS"), ("linear_sb_tuned.h", "UPP_SB_TUNED_CONFIGS")):
        m = re.search(r'#define %s\(X\) (.*)\n' % macro, open(os.path.join(csrc, f)).read())
        for t in re.findall(r'X\(([^)]*)\)', m.group(1).replace('UPP_SB_NST44', '3')):
            a, b, c, d, e = (int(v) for v in t.split(','))
            out.append(0x400000 + a * 65536 + b * 4096 + c * 256 + d * 16 + e)
    return out


SB_CONFIGS = _compiled_tiles()
assert len(SB_CONFIGS) >= 10 and len(set(SB_CONFIGS)) == len(SB_CONFIGS)
TOKENS = [2400, 2080, 2048, 1120]
LAYERS = [("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536), ("dqkv", 384, 1152)]


def _operands(M, N, K, seed=0, spread=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) * K ** -0.5
    if spread:                       # a wide range of magnitudes inside every row
        a = a * torch.exp2(torch.randint(-spread, spread + 1, (M, K), device='cuda', generator=g).float())
        w = w * torch.exp2(torch.randint(-spread, spread + 1, (N, K), device='cuda', generator=g).float())
    b = torch.randn(N, device='cuda', generator=g)
    w._upp_persistent = True             # (a plain tensor standing for a frozen weight: ops.PLANES may keep its plane image)
    return a, w, b


def _sb(a, w, bias=None, epilogue=ops.LIN_NONE, aux=None, tile=0):
    """upp_linear_sb_f32 with a forced tile (ops.linear_f32(..., frozen=True) takes the library's choice)."""
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, device='cuda')
    d = torch.empty(M, N, device='cuda') if epilogue == ops.LIN_BIAS_GELU_D else aux
    ops._call(a.device, "upp_linear_sb_f32", _abi.ptr(a), a.stride(0), _abi.ptr(ops.PLANES.get(w)), _abi.ptr(bias), _abi.ptr(out), N,
              _abi.ptr(d), N, M, N, K, int(epilogue), int(tile))
    return (out, d) if epilogue == ops.LIN_BIAS_GELU_D else out


def _planes_to_terms(planes, N, K):
    """The plane image [block][k-stage][plane][granule][row][8 bf16] -> (3, N, K) f32."""
    nb, ks = (N + 31) // 32, (K + 31) // 32
    t = planes.view(torch.int16).view(nb, ks, 3, 4, 32, 8).permute(2, 0, 4, 1, 3, 5).reshape(3, nb * 32, ks * 32)
    return (t.to(torch.int32) << 16).view(torch.float32)


@pytest.mark.parametrize("shape", [(384, 384), (1536, 384), (40, 128), (50, 100), (33, 32)])
def test_prep_writes_the_exact_three_way_split_in_the_kernel_image(shape):
    N, K = shape
    w = torch.randn(N, K, device='cuda') * torch.exp2(torch.randint(-12, 13, (N, K), device='cuda').float())
    planes = ops.PLANES.get(w)
    assert planes.numel() == _abi.load().upp_linear_sb_planes_bytes(N, K) == ((N + 31) // 32) * ((K + 31) // 32) * 6144
    terms = _planes_to_terms(planes, N, K).cpu().numpy()
    want = np.stack(oracle.split3_bf16(w.cpu().numpy()))
    assert np.array_equal(terms[:, :N, :K], want)                                  # bit for bit the restated split
    assert not terms[:, N:, :].any() and not terms[:, :, K:].any()                 # zero beyond N and K
    assert np.array_equal(terms[:, :N, :K].astype(np.float64).sum(0), w.cpu().numpy().astype(np.float64))   # and it is exact


@pytest.mark.parametrize("cfg", SB_CONFIGS, ids=[hex(c) for c in SB_CONFIGS])
@pytest.mark.parametrize("shape", [(75, 384, 384), (2400, 1152, 384), (333, 96, 256), (1, 40, 128), (129, 1536, 512), (2080, 384, 1536), (31, 36, 64)])
def test_every_compiled_tile_small_integers_exactly(cfg, shape):
    """Integer operands (|.| <= 16): every partial sum is an integer below 2^24, so ANY correct summation order gives the exact
    product -- a layout test (rows, columns, k-groups, edge tiles, the asymmetric weight) that no tolerance can blur."""
    M, N, K = shape
    if K % (32 * ((cfg >> 4) & 15)) or K // (32 * ((cfg >> 4) & 15)) < (cfg & 15):
        pytest.skip("the wave groups' k-stages do not divide K, or fewer k-stages than LDS stages")
    g = torch.Generator(device='cuda').manual_seed(cfg)
    a = torch.randint(-16, 17, (M, K), device='cuda', generator=g).float()
    w = torch.randint(-16, 17, (N, K), device='cuda', generator=g).float()
    got = _sb(a, w, tile=cfg)             # (w is a temporary to ops.PLANES: split at this use, not cached)
    assert torch.equal(got.double(), a.double() @ w.double().t())


@pytest.mark.parametrize("M", TOKENS)
@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
@pytest.mark.parametrize("spread", [0, 8])
def test_block_shapes_are_as_accurate_as_the_exact_f32_kernel(M, layer, spread):
    _, N, K = layer
    a, w, b = _operands(M, N, K, seed=M + N, spread=spread)
    assert ops.linear_sb_tile(M, N, K) > 0
    got = ops.linear_f32(a, w, b, ops.LIN_BIAS, frozen=True)
    f32 = ops.linear_f32(a, w, b, ops.LIN_BIAS)
    exact = a.double() @ w.double().t() + b.double()
    bound = a.abs().double() @ w.abs().double().t() + b.abs().double()            # sum over k of |a w|: the scale every rounding refers to
    e_sb = ((got.double() - exact).abs() / bound).max().item()
    e_f32 = ((f32.double() - exact).abs() / bound).max().item()
    # measured on MI355X over these 40 cases: e_sb 1.3e-7 ... 3.4e-7, e_f32 0.9e-7 ... 4.1e-7 -- the f32 accumulation of either kernel
    assert e_sb <= 1.5 * e_f32 + 1e-7, (e_sb, e_f32)          # (spread = 8: both 0.6e-6 ... 1.5e-6)
    # north_star's bar, with a decade to spare: 1e-6 of the output scale
    assert (got.double() - exact).abs().max().item() <= 1e-6 * exact.abs().max().item()
    # and the kernel adds only f32 accumulation rounding to the restated arithmetic (oracle.linear_split: the six products, exact sums)
    if M <= 1200:
        split = torch.from_numpy(oracle.linear_split(a.cpu().numpy(), w.cpu().numpy(), b.cpu().numpy())).cuda()
        assert ((got.double() - split).abs() / bound).max().item() <= 1.5 * e_f32 + 1e-7


@pytest.mark.parametrize("cfg", SB_CONFIGS, ids=[hex(c) for c in SB_CONFIGS])
def test_epilogues_match_the_exact_f32_kernel(cfg):
    M, N, K = 333, 224, 256
    if K // (32 * ((cfg >> 4) & 15)) < (cfg & 15):       # (fewer k-stages per wave group than LDS stages: a deeper problem for this shape)
        K = 512
    a, w, b = _operands(M, N, K, seed=cfg)
    tol = dict(rtol=1e-5, atol=5e-6)
    torch.testing.assert_close(_sb(a, w, b, ops.LIN_BIAS, tile=cfg), ops.linear_f32(a, w, b, ops.LIN_BIAS), **tol)
    torch.testing.assert_close(_sb(a, w, b, ops.LIN_BIAS_RELU, tile=cfg), ops.linear_f32(a, w, b, ops.LIN_BIAS_RELU), **tol)
    torch.testing.assert_close(_sb(a, w, b, ops.LIN_BIAS_GELU, tile=cfg), ops.linear_f32(a, w, b, ops.LIN_BIAS_GELU), **tol)
    h, d = _sb(a, w, b, ops.LIN_BIAS_GELU_D, tile=cfg)
    h0, d0 = ops.linear_f32(a, w, b, ops.LIN_BIAS_GELU_D)
    torch.testing.assert_close(h, h0, **tol)
    torch.testing.assert_close(d, d0, **tol)
    fac = torch.randn(M, N, device='cuda')
    torch.testing.assert_close(_sb(a, w, None, ops.LIN_MUL, aux=fac, tile=cfg), ops.linear_f32(a, w, None, ops.LIN_MUL, aux=fac), **tol)


def test_shapes_the_kernel_does_not_take_fall_back_to_the_exact_f32_kernel():
    assert ops.linear_sb_tile(2400, 384, 100) == 0          # K % 32
    assert ops.linear_sb_tile(2400, 384, 32) == 0           # fewer k-stages than any compiled tile's LDS stages
    t64 = ops.linear_sb_tile(2400, 384, 64)                 # (round 6: two k-stages fill a two-stage tile -- the fitted model may take one)
    assert t64 == 0 or (64 // (32 * ((t64 >> 4) & 15)) >= (t64 & 15))
    if t64:
        a, w, b = _operands(2400, 384, 64)
        torch.testing.assert_close(ops.linear_f32(a, w, b, ops.LIN_BIAS, frozen=True), ops.linear_f32(a, w, b, ops.LIN_BIAS), rtol=1e-5, atol=5e-6)
    a, w, _ = _operands(64, 48, 100)
    assert torch.equal(ops.linear_f32(a, w, frozen=True), ops.linear_f32(a, w))
    lib = _abi.load()
    out = torch.empty(64, 48, device='cuda')
    pl = torch.empty(int(lib.upp_linear_sb_planes_bytes(48, 100)), dtype=torch.uint8, device='cuda')
    assert lib.upp_linear_sb_f32(_abi.ptr(a), 100, _abi.ptr(pl), None, _abi.ptr(out), 48, None, 0, 64, 48, 100, 0, 0, None) != 0


def test_plane_cache_follows_the_weight_and_its_cached_transpose():
    lin = torch.nn.Linear(384, 1152, bias=False).cuda().requires_grad_(False)
    x = torch.randn(2400, 384, device='cuda', requires_grad=True)
    y = HF.linear(x, lin.weight)
    y.sum().backward()
    g0 = x.grad.clone()
    torch.testing.assert_close(y, x.detach() @ lin.weight.t(), rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(g0, torch.ones_like(y) @ lin.weight, rtol=2e-6, atol=2e-5)
    with torch.no_grad():
        lin.weight.mul_(-2.0)                              # a new version of the frozen weight (load_state_dict does the same)
    x.grad = None
    y2 = HF.linear(x, lin.weight)
    y2.sum().backward()
    torch.testing.assert_close(y2, -2.0 * y, rtol=1e-5, atol=5e-6)
    torch.testing.assert_close(x.grad, -2.0 * g0, rtol=1e-5, atol=5e-5)


def test_a_write_through_data_needs_refresh_caches_and_a_stale_image_is_refused_under_capture():
    """ops._WeightPlanes CONTRACT: validity is keyed on torch's version counter.  `.data` writes bypass it (the image stays as it was --
    that is the documented blind spot), functional.refresh_caches() brings every derived copy up to date in place, and a stale image met
    while a stream is capturing raises instead of being baked into the graph (round-4 advisor)."""
    if not ops.SPLIT_BF16:
        pytest.skip("UPP_SPLIT_BF16=0: no plane images are made")
    lin = torch.nn.Linear(384, 384, bias=False).cuda().requires_grad_(False)
    x = torch.randn(2400, 384, device='cuda', requires_grad=True)
    y = HF.linear(x, lin.weight)
    y.sum().backward()
    g0 = x.grad.clone()
    image = ops.PLANES.get(lin.weight)
    lin.weight.data.mul_(3.0)                                  # no version bump
    assert ops.PLANES.get(lin.weight) is image
    HF.refresh_caches()
    x.grad = None
    y3 = HF.linear(x, lin.weight)
    y3.sum().backward()
    assert ops.PLANES.get(lin.weight) is image                 # refreshed IN PLACE: captured graphs keep reading this address
    torch.testing.assert_close(y3, 3.0 * y, rtol=1e-5, atol=5e-6)
    torch.testing.assert_close(x.grad, 3.0 * g0, rtol=1e-5, atol=5e-5)
    with torch.no_grad():
        lin.weight.mul_(0.5)                                   # version moved, image now stale
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    raised = False
    with torch.cuda.stream(side):
        try:
            with torch.cuda.graph(graph, stream=side):
                try:
                    ops.linear_f32(x.detach(), lin.weight, frozen=True)
                except RuntimeError as e:
                    raised = "refresh_caches" in str(e)
        except RuntimeError:
            pass                                               # (an empty capture may itself complain on some builds)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert raised
    y4 = HF.linear(x.detach(), lin.weight)                     # eager: the version check refreshes the image
    torch.testing.assert_close(y4, 1.5 * y.detach(), rtol=1e-5, atol=5e-6)


def test_non_finite_operands_stay_visible():
    """include/upp_hip.h "Value range": a non-finite operand (weight or activation) makes every output it reaches non-finite -- exactly
    the outputs that are non-finite on the exact-f32 kernel (NaN where that kernel has +-inf: the six-product sum meets inf - inf) -- and
    nothing else is disturbed (round-4 advisor: overflow must stay visible)."""
    M, N, K = 256, 128, 128
    g = torch.Generator(device='cuda').manual_seed(3)
    assert ops.linear_sb_tile(M, N, K) != 0
    for poison_w in (True, False):
        a = torch.randn(M, K, device='cuda', generator=g)
        w = torch.randn(N, K, device='cuda', generator=g)
        w._upp_persistent = True
        if poison_w:
            w[1, 0] = float('inf'); w[2, 5] = float('-inf'); w[3, 7] = float('nan'); w[4, 9] = 3.3e38
        else:
            a[3, 0] = float('inf'); a[5, 1] = float('-inf'); a[7, 2] = float('nan')
        got, ref = ops.linear_f32(a, w, frozen=True).cpu(), ops.linear_f32(a, w).cpu()
        assert torch.equal(torch.isfinite(got), torch.isfinite(ref))
        fin = torch.isfinite(ref)
        assert int((~fin).sum()) >= 3 * (M if poison_w else N)
        torch.testing.assert_close(got[fin], ref[fin], rtol=2e-5, atol=1e-5 * float(ref[fin].abs().max()))


def test_a_trainable_weight_stays_on_the_exact_f32_kernel():
    if not ops.SPLIT_BF16:
        pytest.skip("UPP_SPLIT_BF16=0: the split kernel is switched off")
    w = torch.randn(384, 384, device='cuda', requires_grad=True)
    x = torch.randn(2400, 384, device='cuda')
    with ops.time_linear_calls() as scope:
        HF.linear(x, w)
        HF.linear(x, w.detach())
    torch.cuda.synchronize()
    assert [c[6] for c in scope.calls] == [0, ops.linear_sb_tile(2400, 384, 384)]


def test_plane_images_are_cached_per_owner_object_and_never_for_temporaries():
    lin = torch.nn.Linear(384, 384, bias=False).cuda().requires_grad_(False)
    p1 = ops.PLANES.get(lin.weight)
    assert ops.PLANES.get(lin.weight) is p1                                   # a parameter: one image, kept
    tmp = lin.weight.detach().clone()
    q1, q2 = ops.PLANES.get(tmp), ops.PLANES.get(tmp)
    assert q1 is not q2 and torch.equal(q1, q2) and torch.equal(q1, p1)        # a temporary: split at every use, into its own buffer
    key = (lin.weight.data_ptr(), tuple(lin.weight.shape), tuple(lin.weight.stride()))
    other = torch.nn.Parameter(torch.randn(384, 384, device='cuda'), requires_grad=False)
    ops.PLANES.entries[(other.data_ptr(), tuple(other.shape), tuple(other.stride()))] = ops.PLANES.entries[key]   # "another tensor landed on this address"
    p_other = ops.PLANES.get(other)
    assert p_other is not p1                                                   # the entry was made for another owner object: not served
    terms = _planes_to_terms(p_other, 384, 384).double().sum(0)
    assert torch.equal(terms, other.detach().double())


def test_transposed_and_batched_prep_equal_the_plain_split():
    ws = [torch.randn(384, 1536, device='cuda'), torch.randn(1152, 384, device='cuda'), torch.randn(50, 100, device='cuda')]
    for w in ws:
        direct = ops._WeightPlanes._split(w.t().contiguous())                   # planes of w^T from an explicit copy
        assert torch.equal(ops._WeightPlanes._split(w, None, True), direct)       # ... and read transposed from w itself
    ps = [torch.nn.Parameter(w) for w in ws]
    ops.PLANES.trainable.clear()
    imgs = [ops.PLANES.get_trainable(p, transposed=(i == 1)) for i, p in enumerate(ps)]
    with torch.no_grad():
        for p in ps:
            p.mul_(1.7).add_(0.3)                                                 # "the optimizer stepped"
    ops.PLANES.refresh_trainable()                                                # one launch for all three
    for i, p in enumerate(ps):
        assert torch.equal(imgs[i], ops._WeightPlanes._split(p.detach(), None, i == 1))
    ops.PLANES.trainable.clear()


@pytest.mark.parametrize("shape", [(4096, 1024, 512), (65536, 512, 256), (8192, 256, 1536)])
def test_tall_matrices_run_in_several_rounds(shape):
    M, N, K = shape
    assert ops.linear_sb_tile(M, N, K) > 0
    g = torch.Generator(device='cuda').manual_seed(M)
    a = torch.randint(-8, 9, (M, K), device='cuda', generator=g).float()
    w = torch.randint(-8, 9, (N, K), device='cuda', generator=g).float()
    w._upp_persistent = True
    got = ops.linear_f32(a, w, frozen=True)
    assert torch.equal(got.double(), a.double() @ w.double().t())


def test_group_bias_epilogue_equals_product_plus_broadcast():
    M, N, K, r = 65536, 512, 256, 2048
    a, w, _ = _operands(M, N, K, seed=3)
    gb = torch.randn(M // r, N, device='cuda')
    assert ops.linear_group_bias_usable(M, N, K, r)
    got = ops.linear_group_bias(a, w, gb, r, frozen=True)
    want = ops.linear_f32(a, w, frozen=True).view(M // r, r, N) + gb.unsqueeze(1)
    assert torch.equal(got, want.view(M, N))                                      # same kernel, same sums: the bias is one f32 add either way
    torch.testing.assert_close(got, ops.linear_group_bias(a, w, gb, r), rtol=1e-5, atol=5e-6)


def test_trainable_weights_take_the_split_kernel_inside_a_step_driver_only():
    if not ops.SPLIT_BF16:
        pytest.skip("UPP_SPLIT_BF16=0: the split kernel is switched off")
    lin = torch.nn.Linear(384, 1536).cuda()
    x = torch.randn(2400, 384, device='cuda', requires_grad=True)
    with ops.time_linear_calls() as scope:
        HF.linear(x, lin.weight, lin.bias).sum().backward()
    assert [c[6] for c in scope.calls] == [0, 0]                                  # outside a driver: exact-f32 kernel, forward and data gradient
    g_ref = x.grad.clone()
    x.grad = None
    ops.PLANES.managed = True
    try:
        ops.PLANES.refresh_trainable()
        with ops.time_linear_calls() as scope:
            y = HF.linear(x, lin.weight, lin.bias)
            y.sum().backward()
        assert all(c[6] for c in scope.calls) and len(scope.calls) == 2
        torch.testing.assert_close(x.grad, g_ref, rtol=1e-5, atol=1e-4)
        with torch.no_grad():
            lin.weight.mul_(0.5)
        ops.PLANES.refresh_trainable()                                            # what TrainStep does at the start of every step
        torch.testing.assert_close(HF.linear(x, lin.weight, lin.bias) - lin.bias, 0.5 * (y - lin.bias), rtol=1e-5, atol=5e-6)
    finally:
        ops.PLANES.managed = False
        ops.PLANES.trainable.clear()


# ------------------------------------------------------------------ weight gradients on the bf16 pipe (csrc/wgrad_sb.hip)
def _wgrad(pairs, split):
    from upp_hip import ops
    saved = ops.WGRAD_SPLIT_BF16
    ops.WGRAD_SPLIT_BF16 = split
    try:
        return [p.sum(0) if p.shape[0] > 1 else p[0] for p in ops.linear_wgrad_grouped(pairs)]
    finally:
        ops.WGRAD_SPLIT_BF16 = saved


@pytest.mark.parametrize("M,N,K", [(700, 200, 132), (5, 4, 4), (33, 36, 260), (4096, 128, 128), (1031, 384, 96), (20000, 52, 256),
                                   (140000, 256, 260),          # wide tiles (256 x 256), ragged in K
                                   (70000, 132, 380),           # 128 x 384 tiles
                                   (70000, 1152, 136),          # 384 x 128: operands swapped, tile stored transposed
                                   (300, 384, 384)])            # eligible, but not a round of wide units: 128 x 128 tiles
def test_wgrad_sb_is_exact_on_small_integers(M, N, K):
    """Integer operands of a few bits: every bf16 term, product and f32 partial sum is exact, so the transposed fragment reads, the
    operand maps, the row masks of the last step, the clamped columns of partial tiles and the split order are all checked bit for bit."""
    gen = torch.Generator().manual_seed(M + N + K)
    g = torch.randint(-7, 8, (M, N), generator=gen).float().cuda()
    x = torch.randint(-7, 8, (M, K), generator=gen).float().cuda()
    (dw,) = _wgrad([(g, x)], True)
    want = (g.double().t() @ x.double()).float()
    assert torch.equal(dw, want)


def test_wgrad_sb_group_and_strided_operands():
    """A group of three problems in one launch, operands that are column slices of wider matrices (leading dimensions > widths)."""
    gen = torch.Generator().manual_seed(11)
    big = torch.randint(-5, 6, (900, 640), generator=gen).float().cuda()
    pairs = [(big[:, :256], big[:, 256:640]), (big[:513, 4:68], big[:513, 100:228]), (big[:64, 8:12], big[:64, 320:352])]
    got = _wgrad(pairs, True)
    for (g, x), dw in zip(pairs, got):
        assert torch.equal(dw, (g.double().t() @ x.double()).float())


def test_wgrad_sb_error_is_that_of_the_f32_kernel():
    """Random operands with a wide dynamic range: error against float64 in units of sum |g| |x|, beside the exact-f32 kernel's."""
    gen = torch.Generator().manual_seed(5)
    M, N, K = 8192, 256, 384
    g = (torch.randn(M, N, generator=gen) * torch.exp(2 * torch.randn(M, 1, generator=gen))).cuda()
    x = (torch.randn(M, K, generator=gen) * torch.exp(2 * torch.randn(1, K, generator=gen))).cuda()
    (sb,), (f32,) = _wgrad([(g, x)], True), _wgrad([(g, x)], False)
    want = g.double().t() @ x.double()
    scale = g.double().abs().t() @ x.double().abs()
    e_sb, e_f32 = ((sb.double() - want).abs() / scale).max().item(), ((f32.double() - want).abs() / scale).max().item()
    print("wgrad error / sum |g||x|: split bf16 %.2e, exact f32 %.2e" % (e_sb, e_f32))
    assert e_sb <= 6e-7 and e_sb <= 3 * e_f32 + 1e-7


def test_wgrad_sb_special_values_and_determinism():
    gen = torch.Generator().manual_seed(6)
    g, x = torch.randn(3000, 132, generator=gen).cuda(), torch.randn(3000, 200, generator=gen).cuda()
    a, b = _wgrad([(g, x)], True)[0], _wgrad([(g, x)], True)[0]
    assert torch.equal(a, b)
    g[17, 5] = float('inf')
    dw = _wgrad([(g, x)], True)[0]
    assert not torch.isfinite(dw[5]).any() and torch.isfinite(dw[6]).all()
