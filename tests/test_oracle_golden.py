"""Pin the CPU oracle: against the reference's own known answers where they exist
(EMD: extensions/emd/test_emd_loss.py; Chamfer: extensions/chamfer_dist/test.py gradcheck),
and against independent numpy restatements for the unpinned third-party ops (FPS / kNN)."""
import numpy as np
import pytest

import oracle as O


def test_emd_known_answer_reference_test_emd_loss():
    # reference extensions/emd/test_emd_loss.py:7-19: the obvious permutation, cost 0.30 + 0.41
    p1 = np.array([[[1.7, -0.1, 0.1], [0.1, 1.2, 0.3]]], np.float32).repeat(3, 0)
    p2 = np.array([[[0.3, 1.8, 0.2], [1.2, -0.2, 0.3]]], np.float32).repeat(3, 0)
    match = O.emd_approxmatch(p1, p2)
    cost = O.emd_matchcost(p1, p2, match)
    gt = ((p1[0, 0] - p2[0, 1]) ** 2).sum() + ((p1[0, 1] - p2[0, 0]) ** 2).sum()
    np.testing.assert_allclose(cost, [gt] * 3, rtol=1e-6)
    np.testing.assert_allclose(match[0], [[0, 1], [1, 0]], atol=1e-8)
    # loss = d0/2 + 2*d1 + d2/3  (reference :41) -> grads = coef * 2 * (p - q) of the matched pairs
    coef = np.array([0.5, 2.0, 1.0 / 3.0], np.float32)
    g1, g2 = O.emd_matchcost_grad(coef, p1, p2, match)
    for b in range(3):
        np.testing.assert_allclose(g1[b, 0], coef[b] * 2 * (p1[b, 0] - p2[b, 1]), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(g1[b, 1], coef[b] * 2 * (p1[b, 1] - p2[b, 0]), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(g2[b, 1], coef[b] * 2 * (p2[b, 1] - p1[b, 0]), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(g2[b, 0], coef[b] * 2 * (p2[b, 0] - p1[b, 1]), rtol=1e-5, atol=1e-7)


def test_emd_unequal_sizes_mass_conservation():
    rng = np.random.default_rng(1)
    p1 = rng.random((2, 96, 3), dtype=np.float32)
    p2 = rng.random((2, 32, 3), dtype=np.float32)
    match = O.emd_approxmatch(p1, p2)            # (B, 32, 96); multiR = 96/32 = 3
    np.testing.assert_allclose(match.sum(axis=1), 1.0, atol=2e-3)   # every xyz1 point ships ~1 unit
    np.testing.assert_allclose(match.sum(axis=2), 3.0, atol=6e-3)   # every xyz2 point receives ~3


def test_chamfer_matches_numpy_and_first_min_rule():
    rng = np.random.default_rng(0)
    a = rng.random((3, 70, 3), dtype=np.float32)
    b = rng.random((3, 1100, 3), dtype=np.float32)       # > 2 tiles of 512
    b[:, 900] = b[:, 5]                                   # exact duplicate: lower index must win
    d1, d2, i1, i2 = O.chamfer_fwd(a, b)
    dn = ((a[:, :, None, :].astype(np.float64) - b[:, None, :, :]) ** 2).sum(-1)
    np.testing.assert_allclose(d1, dn.min(2), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(d2, dn.min(1), rtol=1e-5, atol=1e-7)
    assert not (i1 == 900).any()
    assert (np.take_along_axis(dn, i1[..., None].astype(np.int64), 2)[..., 0] <= dn.min(2) * (1 + 1e-5) + 1e-7).all()


def test_chamfer_backward_is_gradient_of_forward():
    # restates reference extensions/chamfer_dist/test.py:24-29 (gradcheck on (4,64,3) x (4,128,3))
    rng = np.random.default_rng(2)
    x = rng.random((4, 64, 3)).astype(np.float32)
    y = rng.random((4, 128, 3)).astype(np.float32)
    w1 = rng.random((4, 64)).astype(np.float32)
    w2 = rng.random((4, 128)).astype(np.float32)

    def loss(xx, yy):
        d1, d2, _, _ = O.chamfer_fwd(xx, yy)
        return float((d1.astype(np.float64) * w1).sum() + (d2.astype(np.float64) * w2).sum())

    d1, d2, i1, i2 = O.chamfer_fwd(x, y)
    g1, g2 = O.chamfer_bwd(x, y, i1, i2, w1, w2)
    eps = 1e-3
    for (b, j, c) in [(0, 0, 0), (1, 17, 2), (3, 63, 1)]:
        xp, xm = x.copy(), x.copy()
        xp[b, j, c] += eps; xm[b, j, c] -= eps
        assert abs((loss(xp, y) - loss(xm, y)) / (2 * eps) - g1[b, j, c]) < 2e-2 * max(1.0, abs(g1[b, j, c]))
    for (b, j, c) in [(0, 5, 1), (2, 127, 0)]:
        yp, ym = y.copy(), y.copy()
        yp[b, j, c] += eps; ym[b, j, c] -= eps
        assert abs((loss(x, yp) - loss(x, ym)) / (2 * eps) - g2[b, j, c]) < 2e-2 * max(1.0, abs(g2[b, j, c]))


def _fps_numpy(p, m):
    """Independent restatement without the CUDA thread structure (valid when no exact ties occur)."""
    n = len(p)
    d = np.full(n, 1e10, np.float32)
    mag = (p.astype(np.float32) ** 2).sum(1)
    out = [0]
    for _ in range(1, m):
        q = p[out[-1]]
        dx, dy, dz = (p[:, 0] - q[0]), (p[:, 1] - q[1]), (p[:, 2] - q[2])
        dd = (dx * dx + dy * dy + dz * dz).astype(np.float32)
        d = np.minimum(d, dd)
        c = d.copy()
        c[mag <= 1e-3] = -1
        out.append(int(c.argmax()))
    return np.array(out)


@pytest.mark.parametrize("N,M", [(1024, 64), (1096, 32), (64, 32), (32, 32), (972, 32), (300, 300)])
def test_fps_against_numpy(N, M):
    rng = np.random.default_rng(N)
    x = rng.standard_normal((3, N, 3)).astype(np.float32)
    x /= np.abs(x).max()
    idx = O.fps(x, M)
    assert idx.dtype == np.int32 and idx.shape == (3, M)
    assert (idx[:, 0] == 0).all()
    for b in range(3):
        ref = _fps_numpy(x[b], M)
        # float ties are measure-zero for gaussian data; fma vs mul+add may still flip a near-tie
        assert (idx[b] == ref).mean() > 0.9


def test_fps_skips_points_near_origin_and_repeats_zero_when_nothing_left():
    x = np.zeros((1, 16, 3), np.float32)
    x[0, :, 0] = np.linspace(0.0, 0.03, 16)          # |p|^2 <= 9e-4 <= 1e-3: every point is skipped
    assert (O.fps(x, 5) == 0).all()
    x[0, 7] = [1.0, 0, 0]
    x[0, 9] = [-1.0, 0, 0]
    idx = O.fps(x, 4)[0]
    assert idx[0] == 0 and set(idx[1:3]) == {7, 9}
    assert idx[3] in (7, 9)                           # only two candidates exist


def test_fps_tie_rule_follows_the_cuda_block_reduction():
    # 8 points -> T = 8 threads, one point each.  Points 1..7 all at distance 1 from point 0:
    # the tree folds t+4 -> t, t+2 -> t, t+1 -> t keeping the LOWER slot on ties, so thread 0's
    # own candidate (k = 0, d = 0) loses to the first real maximum reachable in bit-reversed order.
    x = np.zeros((1, 8, 3), np.float32)
    x[0, 0] = [5, 5, 5]
    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 0, 0]], np.float32)
    x[0, 1:] = x[0, 0] + dirs
    idx = O.fps(x, 2)[0]
    # candidates 1..7 tie at d = 1; bit-reversed thread order is 0,4,2,6,1,5,3,7 -> thread 4 wins
    assert idx[1] == 4
    assert O.fps_block_size(8) == 8 and O.fps_block_size(1000) == 512 and O.fps_block_size(5) == 4


@pytest.mark.parametrize("N,Q,K", [(1024, 64, 32), (64, 32, 8), (32, 32, 16), (100, 7, 100), (5, 3, 1)])
def test_knn_against_numpy_stable_argsort(N, Q, K):
    rng = np.random.default_rng(N + K)
    ref = rng.random((2, N, 3), dtype=np.float32)
    qry = ref[:, :Q].copy() if Q <= N else rng.random((2, Q, 3), dtype=np.float32)
    d, i = O.knn(ref, qry, K)
    assert i.dtype == np.int64 and d.dtype == np.float32
    dx = ref[:, None, :, 0] - qry[:, :, None, 0]
    dy = ref[:, None, :, 1] - qry[:, :, None, 1]
    dz = ref[:, None, :, 2] - qry[:, :, None, 2]
    dn = (dx * dx + dy * dy + dz * dz)
    want = np.argsort(dn, axis=-1, kind='stable')[..., :K]
    assert (want == i).mean() > 0.995      # fma vs mul+add can reorder a near-tie
    np.testing.assert_allclose(d, np.sqrt(np.take_along_axis(dn, i, -1)), rtol=1e-5, atol=1e-7)
    assert (np.diff(d, axis=-1) >= 0).all()


def test_knn_ties_keep_lower_index_first():
    ref = np.zeros((1, 12, 3), np.float32)
    ref[0, :, 0] = [3, 1, 1, 2, 1, 5, 0, 0, 2, 1, 9, 0]
    qry = np.zeros((1, 1, 3), np.float32)
    _, i = O.knn(ref, qry, 8)
    assert i[0, 0].tolist() == [6, 7, 11, 1, 2, 4, 9, 3]
    with pytest.raises(ValueError):
        O.knn(ref, qry, 13)


def test_gather_and_group_roundtrip():
    rng = np.random.default_rng(4)
    xyz = rng.random((2, 50, 3), dtype=np.float32)
    idx = O.fps(xyz, 10)
    feat = np.ascontiguousarray(xyz.transpose(0, 2, 1))
    out = O.gather(feat, idx)
    np.testing.assert_array_equal(out.transpose(0, 2, 1), np.take_along_axis(xyz, idx[..., None].astype(np.int64), 1))
    g = O.gather_grad(np.ones_like(out), idx, 50)
    assert g.sum() == out.size and g.max() == 1
    _, kidx = O.knn(xyz, out.transpose(0, 2, 1).copy(), 4)
    nb = O.group(xyz, out.transpose(0, 2, 1).copy(), kidx)
    np.testing.assert_array_equal(nb[:, :, 0], 0)     # nearest neighbour of a sampled centre is itself


def test_fps_and_knn_reproduce_the_references_own_numpy_and_torch_statements(golden):
    """tests/golden/ref_ops.npz holds what the reference's plain-numpy farthest_point_sample (datasets/ModelNetDataset.py:29-49,
    start index 0) and torch knn_point (models/Transformer_utils.py:17-29) return on seeded clouds without exact ties and
    without points inside the CUDA kernel's |p|^2 <= 1e-3 skip radius: the oracle must give the same index sequence / the
    same neighbour sets."""
    import _seeded
    g = golden["ref_ops"]
    n = 0
    for name, pts, M, Q, k in _seeded.ref_ops_cases(g):
        p = pts.numpy()
        idx = O.fps(p, M)
        np.testing.assert_array_equal(idx, g[name + "/fps"])
        if Q:
            centers = np.stack([p[b][idx[b, :Q]] for b in range(p.shape[0])])
            _, nb = O.knn(p, centers, k, want_dist=False)
            np.testing.assert_array_equal(np.sort(nb, axis=-1), g[name + "/knn"])
        n += 1
    assert n == 6


def test_linear_oracle_is_a_product_and_its_order_is_what_it_says():
    """oracle.linear_f32 (the CPU restatement upp_linear_f32 is checked against bit for bit on the GPU): equals the f64 product
    to f32 accuracy for every (KS, KC), and equals a direct numpy evaluation of the documented fmaf order."""
    import oracle as O
    rng = np.random.default_rng(3)
    a = rng.standard_normal((37, 256)).astype(np.float32)
    w = (rng.standard_normal((24, 256)) / 16).astype(np.float32)
    b = rng.standard_normal(24).astype(np.float32)
    exact = a.astype(np.float64) @ w.astype(np.float64).T
    for ks, kc in ((1, 1), (1, 2), (2, 1), (4, 1), (2, 2)):
        c = O.linear_f32(a, w, ks=ks, kc=kc)
        np.testing.assert_allclose(c, exact, rtol=0, atol=2e-6 * np.abs(exact).max())
        np.testing.assert_array_equal(O.linear_f32(a, w, bias=b, ks=ks, kc=kc, epilogue=1), c + b)
    # the documented order for KS = 2, KC = 1, written out with numpy scalars (fma emulated in f64: exact for f32 operands,
    # one rounding per product as fmaf does)
    def fmaf(x, y, z):
        return np.float32(np.float64(x) * np.float64(y) + np.float64(z))
    m, n = 5, 7
    total = np.float32(0.0)
    for g in range(2):
        acc = np.float32(0.0)
        for c0 in range(0, 256, 64):
            k0 = c0 + g * 32
            for i in range(4):
                for j in range(4):
                    acc = fmaf(a[m, k0 + 8 * i + j], w[n, k0 + 8 * i + j], acc)
                    acc = fmaf(a[m, k0 + 8 * i + 4 + j], w[n, k0 + 8 * i + 4 + j], acc)
        total = np.float32(total + acc)
    assert O.linear_f32(a, w, ks=2, kc=1)[m, n] == total


def test_weight_gradient_oracle_is_a_product_and_its_order_is_what_it_says():
    """oracle.linear_wgrad (what upp_linear_wgrad_grouped_f32 is checked against bit for bit on the GPU): runs of `rows` rows, each entry
    one ascending-row fmaf chain; the runs add up to the f64 product to f32 accuracy."""
    rng = np.random.default_rng(5)
    M, N, K, rows = 150, 12, 20, 64
    g = rng.standard_normal((M, N)).astype(np.float32)
    x = rng.standard_normal((M, K)).astype(np.float32)
    part = O.linear_wgrad(g, x, rows)
    assert part.shape == (3, N, K)
    ref = g.astype(np.float64).T @ x.astype(np.float64)
    np.testing.assert_allclose(part.astype(np.float64).sum(0), ref, rtol=0, atol=4e-6 * np.abs(ref).max())
    import math
    for s, n, k in ((0, 0, 0), (1, 5, 7), (2, 11, 19)):
        acc = np.float32(0.0)
        for m in range(s * rows, min(M, (s + 1) * rows)):
            acc = np.float32(math.fma(float(g[m, n]), float(x[m, k]), float(acc))) if hasattr(math, "fma") else np.float32(np.float64(g[m, n]) * np.float64(x[m, k]) + np.float64(acc))
        assert part[s, n, k] == acc


def test_small_k_linear_oracle_is_the_plain_f32_product():
    import oracle as O
    rng = np.random.default_rng(3)
    x = rng.standard_normal((50, 59)).astype(np.float32)
    w = rng.standard_normal((32, 59)).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    np.testing.assert_allclose(O.linear_smallk(x, w, b), ref, rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(O.linear_smallk(x, w, b, 1), np.maximum(O.linear_smallk(x, w, b), 0))
