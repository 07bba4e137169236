"""ABI v11 vector passes of the meta-gradient (csrc/l2o_vecops.h: l2o_suffix_sums, l2o_colsum, l2o_lincomb,
l2o_rnnprop_input_adjoint) against float64 / the host formulas they replaced in open_l2o_amd/meta.py (round 3 ran them
as torch tensor arithmetic).  The branches that use them -- generic `layers` BPTT, the Linear-only net, second
derivatives -- keep their own end-to-end tests (test_generic_net, test_second_derivatives, test_rnnprop_gradient)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from open_l2o_amd._engine import HipEngine
    return HipEngine()


@pytest.mark.parametrize("T,n", [(1, 7), (20, 16384), (100, 1000), (5, 300001)])
def test_suffix_sums(eng, T, n):
    rng = np.random.default_rng(T * 131 + n)
    gs = [rng.standard_normal(n).astype(np.float32) for _ in range(T)]
    gf = rng.standard_normal(n).astype(np.float32)
    out = eng.empty(T, n)
    eng.suffix_sums([eng.tensor(g) for g in gs], eng.tensor(gf), out)
    want = np.empty((T, n), np.float32)
    acc = gf.copy()
    for t in reversed(range(T)):                        # the host loop of round 3, fp32, same order: bit-identical
        want[t] = acc
        acc = acc + gs[t]
    assert np.array_equal(eng.to_numpy(out), want)


@pytest.mark.parametrize("shape", [(1, 1), (16384, 80), (100000, 1), (777, 256), (3, 128, 128), (128, 2, 2)])
def test_colsum(eng, shape):
    rng = np.random.default_rng(sum(shape))
    A = rng.standard_normal(shape).astype(np.float32)
    want = A.astype(np.float64).sum(axis=-2)
    got = eng.to_numpy(eng.colsum(eng.tensor(A)))
    scale = np.abs(A).astype(np.float64).sum(axis=-2).max()
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-7 * scale + 1e-30
    again = eng.to_numpy(eng.colsum(eng.tensor(A)))     # fixed summation order: bit-reproducible
    assert np.array_equal(got, again)
    if len(shape) == 2:                                  # accumulate
        out = eng.tensor(np.ones(shape[1], np.float32))
        eng.colsum(eng.tensor(A), out=out, accumulate=True)
        assert np.abs(eng.to_numpy(out) - (1.0 + want)).max() <= 2e-7 * scale + 1e-6


def test_lincomb_and_aliasing(eng):
    rng = np.random.default_rng(5)
    a, b, c = (rng.standard_normal(4099).astype(np.float32) for _ in range(3))
    ta, tb, tc = eng.tensor(a), eng.tensor(b), eng.tensor(c)
    out = eng.empty(4099)
    eng.lincomb(out, ta, 2.0, tb, -1.0, tc, 0.25)
    np.testing.assert_allclose(eng.to_numpy(out), 2.0 * a - b + 0.25 * c, rtol=1e-6, atol=1e-6)
    eng.lincomb(ta, ta, 1.0, tb, 1.0)                    # in place: acc += val
    assert np.array_equal(eng.to_numpy(ta), a + b)
    eng.lincomb(out, tc, 3.0)
    assert np.array_equal(eng.to_numpy(out), np.float32(3.0) * c)


@pytest.mark.parametrize("k", [1, 7, 150])
def test_rnnprop_input_adjoint_vs_float64(eng, k):
    """The formulas of DM/meta_rnnprop_train.py:380-388 differentiated by hand (see the kernel's comment), in float64."""
    rng = np.random.default_rng(k)
    n, H, KB = 5000, 20, 8 * 20 + 1 + 20
    Bm = rng.standard_normal((n + 3, KB)).astype(np.float32)
    wfc = (rng.standard_normal((2, H)) * 0.3).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    m = (rng.standard_normal(n) * 0.3).astype(np.float32)
    v = (rng.random(n) * 0.2).astype(np.float32)
    v[:10] = 0.0                                         # sqrt(v^) = 0: the guarded branch
    dm0, dv0 = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    b1 = b2 = float(np.float32(0.95))
    tdm, tdv, tdg = eng.tensor(dm0), eng.tensor(dv0), eng.empty(n)
    eng.rnnprop_input_adjoint(eng.tensor(Bm), 8 * H + 1, H, eng.tensor(wfc), eng.tensor(g), eng.tensor(m), eng.tensor(v),
                              b1 ** k, b2 ** k, b1, b2, tdm, tdv, tdg)
    f8 = np.float64
    du = Bm[:n, 8 * H + 1:8 * H + 1 + H].astype(f8)
    a0, a1 = du @ wfc[0].astype(f8), du @ wfc[1].astype(f8)
    om1, om2 = 1.0 - b1 ** k, 1.0 - b2 ** k
    m_hat, sq = m.astype(f8) / om1, np.sqrt(v.astype(f8) / om2)
    den = sq + 1e-8
    d_den = -(a0 * m_hat + a1 * g) / den ** 2
    d_vhat = np.where(sq > 0, d_den * 0.5 / np.maximum(sq, 1e-30), 0.0)
    dm = a0 / den / om1 + dm0
    dv = d_vhat / om2 + dv0
    dg = a1 / den + dm * (1.0 - b1) + dv * 2.0 * (1.0 - b2) * g
    # fp32 against float64, term by term: every output is a sum of terms of mixed sign, so the error is bounded
    # relative to the sum of the terms' MAGNITUDES (rows with v = 0 have den = 1e-8 and adjoints of 1e8 .. 1e16)
    mag_dm = np.abs(a0 / den / om1) + np.abs(dm0)
    mag_dv = np.where(sq > 0, (np.abs(a0 * m_hat) + np.abs(a1 * g)) / den ** 2 * 0.5 / np.maximum(sq, 1e-30), 0.0) / om2 + np.abs(dv0)
    mag_dg = np.abs(a1 / den) + mag_dm * (1.0 - b1) + mag_dv * 2.0 * (1.0 - b2) * np.abs(g)
    for got, want, mag in ((tdg, dg, mag_dg), (tdm, dm * b1, mag_dm), (tdv, dv * b2, mag_dv)):
        got = eng.to_numpy(got).astype(f8)
        assert np.all(np.isfinite(got))
        assert np.all(np.abs(got - want) <= 2e-5 * mag + 1e-30), float(np.max(np.abs(got - want) / mag))
