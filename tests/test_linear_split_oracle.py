"""The arithmetic of upp_linear_sb_f32 (csrc/linear_sb.hip) on the CPU: the exact three-way bf16 split of f32 operands and the six-product
sum (oracle.split3_bf16 / oracle.linear_split), against float64.  Stands for nn.Linear of the reference
(models/Point_MAE_pretask_dev.py:153-196).  No GPU."""
import numpy as np
import pytest

import oracle


def _cases():
    rng = np.random.default_rng(7)
    yield "normal", rng.standard_normal(4096).astype(np.float32)
    yield "wide range", (rng.standard_normal(4096) * np.exp2(rng.integers(-60, 60, 4096))).astype(np.float32)
    yield "powers of two and neighbours", np.concatenate([np.exp2(np.arange(-20, 20)), np.nextafter(np.exp2(np.arange(-20, 20)).astype(np.float32), np.float32(9e9)),
                                                        -np.nextafter(np.exp2(np.arange(-20, 20)).astype(np.float32), np.float32(0))]).astype(np.float32)
    yield "all 24 significand bits", (np.float32(1) + np.arange(1, 1 << 12, dtype=np.float32) * np.float32(2.0 ** -23) * 2049).astype(np.float32)
    yield "zeros and small", np.array([0.0, -0.0, 1e-30, -3e-33, 2.0 ** -110], np.float32)


@pytest.mark.parametrize("name,x", list(_cases()), ids=[c[0] for c in _cases()])
def test_three_bf16_terms_hold_an_f32_exactly(name, x):
    x1, x2, x3 = oracle.split3_bf16(x)
    for t in (x1, x2, x3):                                   # every term is a bf16 value: low 16 bits clear
        assert not (t.view(np.uint32) & 0xFFFF).any()
    assert np.array_equal(x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64), x.astype(np.float64))
    nz = x != 0
    assert (np.abs(x2[nz]) <= np.abs(x[nz]) * 2.0 ** -8).all() and (np.abs(x3[nz]) <= np.abs(x[nz]) * 2.0 ** -16).all()


@pytest.mark.parametrize("shape", [(64, 48, 384), (33, 20, 1536), (7, 5, 64)])
def test_six_products_are_an_f32_accurate_product(shape):
    M, N, K = shape
    rng = np.random.default_rng(M * N)
    a = (rng.standard_normal((M, K)) * np.exp2(rng.integers(-6, 6, (M, K)))).astype(np.float32)
    w = (rng.standard_normal((N, K)) * K ** -0.5).astype(np.float32)
    exact = a.astype(np.float64) @ w.astype(np.float64).T
    bound = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T          # sum over k of |a w|
    got = oracle.linear_split(a, w)
    # the three dropped terms are each below 2^-24 |a w| (a2 w3, a3 w2) or 2^-32 (a3 w3): one f32 rounding of the product
    assert (np.abs(got - exact) <= 2.0 ** -24 * bound).all()
    # an f32 fmaf chain (the exact-f32 kernel, oracle_linear_f32) is allowed K such roundings; it typically lands at ~sqrt(K) 2^-25
    chain = oracle.linear_f32(a, w).astype(np.float64)
    assert np.abs(got - exact).max() <= np.abs(chain - exact).max()


def test_below_the_bf16_subnormal_grid_the_split_is_off_by_less_than_2_to_minus_133():
    x = np.array([1e-38, -3e-38, 1.5e-40, 2.0 ** -120, -1e-36], np.float32)
    x1, x2, x3 = oracle.split3_bf16(x)
    assert (np.abs(x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64) - x.astype(np.float64)) <= 2.0 ** -133).all()


def test_small_integers_come_out_exact():
    rng = np.random.default_rng(3)
    a = rng.integers(-16, 17, (40, 128)).astype(np.float32)
    w = rng.integers(-16, 17, (24, 128)).astype(np.float32)
    assert np.array_equal(oracle.linear_split(a, w), a.astype(np.float64) @ w.astype(np.float64).T)
