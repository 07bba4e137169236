"""GPU parity tests of the individual HIP kernels, through the C ABI.

Bar (BASELINE.json north_star): fp32, loss trajectories within 1e-5 relative of the
reference CPU path; teacher-forced single steps within ~1e-6 absolute on O(1) state.
"""
import numpy as np
import pytest
import torch

import oracle as O
from helpers import (ORACLE_CFGS, device_problem, lib_option, make_params, make_problem, max_abs, random_state,
                     rel_err, spec_of)
from open_l2o_amd import _abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from open_l2o_amd._engine import HipEngine
    return HipEngine()


def _unpack_state(eng, st, B, D):
    h1, c1, h2, c2 = [eng.to_numpy(t) for t in eng.state_unpack(st, B, D)]
    return ((h1, c1), (h2, c2))


@pytest.mark.parametrize("B,D", [(1, 16), (3, 10), (2, 37), (4, 128)])
def test_state_pack_roundtrip(eng, B, D):
    rng = np.random.default_rng(0)
    arrs = [rng.standard_normal((B * D, 20)).astype(np.float32) for _ in range(4)]
    st = eng.state_pack(*[eng.tensor(a) for a in arrs], B, D)
    assert st.numel() == eng.state_floats(B, D)
    back = [eng.to_numpy(t) for t in eng.state_unpack(st, B, D)]
    for a, b in zip(arrs, back):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
@pytest.mark.parametrize("B,D", [(1, 16), (5, 10), (3, 50), (2, 128)])
def test_lstm_step_teacher_forced(eng, name, B, D):
    """One optimizer step from a random state == oracle net_apply (+ RNNProp moments)."""
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=1)
    rng = np.random.default_rng(2)
    N = B * D
    g = (rng.standard_normal((B, D)) * np.exp(rng.uniform(-6, 1, (B, D)))).astype(np.float32)
    g.flat[0] = 0.0                                   # log(|0| + eps) path
    x0 = rng.standard_normal((B, D)).astype(np.float32)
    state = random_state(cfg, N, seed=3)
    m0 = (rng.standard_normal((B, D)) * 0.1).astype(np.float32)
    v0 = (rng.random((B, D)) * 0.1).astype(np.float32)
    k = 7
    # oracle
    if cfg.kind == "rnnprop":
        dt = np.float32
        m = dt(0.95) * m0 + dt(1 - 0.95) * g
        v = dt(0.95) * v0 + dt(1 - 0.95) * g * g
        mh = m / (dt(1) - np.power(dt(0.95), dt(k)))
        vh = v / (dt(1) - np.power(dt(0.95), dt(k)))
        inputs = (mh / (np.sqrt(vh) + dt(1e-8)), g / (np.sqrt(vh) + dt(1e-8)))
    else:
        inputs, m, v = g, m0, v0
    delta, st_ref = O.net_apply(cfg, params, inputs, state)
    # HIP
    wpack = eng.pack_weights(spec, params)
    st = eng.state_pack(*[eng.tensor(a) for hc in state for a in hc], B, D)
    xd, gd, md, vd = eng.tensor(x0), eng.tensor(g), eng.tensor(m0), eng.tensor(v0)
    pw = float(np.float32(0.95)) ** k
    eng.lstm_step(spec, wpack, gd, md, vd, pw, pw, st, xd, B, D)
    x1 = eng.to_numpy(xd)
    st_hip = _unpack_state(eng, st, B, D)
    e_delta = max_abs(x1 - x0, delta)
    e_state = max(max_abs(st_hip[l][i], st_ref[l][i]) for l in range(2) for i in range(2))
    print("step %s B=%d D=%d: |d delta|=%.3g |d state|=%.3g (|delta|max=%.3g)"
          % (name, B, D, e_delta, e_state, np.abs(delta).max()))
    assert e_state < 3e-6
    # x1 - x0 is formed in fp32 around |x| ~ 3: allow its rounding on top of the kernel error
    assert e_delta < 3e-6 * max(1.0, float(np.abs(delta).max())) + 5e-7
    if cfg.kind == "rnnprop":
        # fma contraction on the GPU vs separately rounded mul/add in NumPy: 1 ulp
        assert max_abs(eng.to_numpy(md), m) < 1.5e-7 and max_abs(eng.to_numpy(vd), v) < 1.5e-7


def test_linear_only_net(eng):
    from open_l2o_amd import _abi
    from open_l2o_amd._engine import ProblemDesc
    cfg = O.NetConfig("cw", (), "identity", None, 1.0, False)
    params = {"linear": {"w": np.full((1, 1), -0.01, np.float32), "b": np.full((1,), -0.01, np.float32)}}
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    x = eng.tensor(np.ones((1, 1), np.float32))
    f = eng.zeros(1)
    g = eng.zeros(1, 1)
    pd = ProblemDesc(_abi.PROB_SIMPLE, 1, 1, 1)
    for _ in range(5):                                  # the meta_test.py:64-69 trajectory
        eng.problem_fg(pd, x, f, g)
        eng.lstm_step(spec, wpack, g, None, None, 0.0, 0.0, None, x, 1, 1)
    eng.problem_fg(pd, x, f, None)
    assert abs(float(eng.to_numpy(x)[0, 0]) - 0.8558813) < 1e-6
    assert abs(float(eng.to_numpy(f)[0]) - 0.7325327) < 5e-5


@pytest.mark.parametrize("kind,B,D,M", [("quadratic", 4, 10, None), ("quadratic", 3, 128, None),
                                        ("quadratic", 2, 300, None), ("lasso", 4, 10, None),
                                        ("lasso", 3, 50, 25), ("lasso", 2, 512, 256),
                                        ("rastrigin", 4, 2, None), ("rastrigin", 3, 100, None),
                                        ("square_cos", 4, 2, None), ("square_cos", 3, 40, None),
                                        # the single-pass kernel k_problem_fg1<NV> (D % 4 == 0, 64 <= D <= 2048)
                                        ("lasso", 2, 64, 37), ("rastrigin", 2, 256, None), ("square_cos", 2, 132, None),
                                        ("lasso", 2, 1024, 70), ("quadratic", 1, 1100, None), ("lasso", 1, 2048, 33)])
def test_problem_fg(eng, kind, B, D, M):
    prob, x0, arrays = make_problem(kind, B, D, seed=4, M=M, stddev=0.5)
    Bg = 2 * B                                         # exercise the global-batch factor
    prob.batch_global = Bg
    rng = np.random.default_rng(5)
    xs = np.exp(rng.uniform(-1, 1, (B, D))).astype(np.float32)
    x2 = x0.reshape(B, D)
    pd = device_problem(eng, arrays, B, D, B_global=Bg, x_scale=xs)
    f = eng.zeros(B)
    g = eng.zeros(B, D)
    eng.problem_fg(pd, eng.tensor(x2), f, g)
    xin = (x2 * xs).reshape(x0.shape)
    f_ref = prob.f_per_problem(xin)
    g_ref = prob.grad(xin).reshape(B, D) * xs
    e_f, e_g = rel_err(eng.to_numpy(f), f_ref), max_abs(eng.to_numpy(g), g_ref) / np.abs(g_ref).max()
    print("fg %s B=%d D=%d: rel f=%.3g, g=%.3g" % (kind, B, D, e_f, e_g))
    assert e_f < 2e-5 and e_g < 5e-6
    fx = eng.zeros(1)
    eng.reduce_fx(f, 1, B, Bg, fx)
    assert rel_err(eng.to_numpy(fx)[0], prob.f(xin)) < 1e-5
    # L2O_OPT_FG_TWO_PASS (l2o_problem.flags): the two-pass form of the same entry point (what a matrix SHARED by the batch
    # takes by default) gives the same numbers as the single pass
    from open_l2o_amd import _abi
    with lib_option(_abi.OPT_FG_TWO_PASS, 1):
        pd2 = device_problem(eng, arrays, B, D, B_global=Bg, x_scale=xs)
        f2, g2 = eng.zeros(B), eng.zeros(B, D)
        eng.problem_fg(pd2, eng.tensor(x2), f2, g2)
    assert rel_err(eng.to_numpy(f2), f_ref) < 2e-5 and max_abs(eng.to_numpy(g2), g_ref) / np.abs(g_ref).max() < 5e-6


@pytest.mark.parametrize("name", ["dm_logsign", "rnnprop"])
def test_lstm_step_multi_equals_separate_steps(eng, name):
    """l2o_cwlstm_step_multi over several (variable, state) panels == one l2o_cwlstm_step per
    panel, bit for bit (problems.mnist shapes: 784x20, 20, 20x10, 10 plus a multi-tile row)."""
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, make_params(cfg, seed=11))
    rng = np.random.default_rng(12)
    shapes = [(784, 20), (1, 20), (20, 10), (1, 10), (3, 37)]
    pw = float(np.float32(0.95)) ** 3

    def fresh():
        r = np.random.default_rng(13)
        segs = []
        for (B, D) in shapes:
            g = eng.tensor((r.standard_normal((B, D)) * 0.3).astype(np.float32))
            m = eng.tensor((r.standard_normal((B, D)) * 0.1).astype(np.float32))
            v = eng.tensor((r.random((B, D)) * 0.1).astype(np.float32))
            x = eng.tensor(r.standard_normal((B, D)).astype(np.float32))
            state = random_state(cfg, B * D, seed=int(r.integers(1 << 30)))
            st = eng.state_pack(*[eng.tensor(a) for hc in state for a in hc], B, D)
            segs.append((g, m, v, st, x, B, D))
        return segs

    a, b = fresh(), fresh()
    eng.lstm_step_multi(spec, wpack, a, pw, pw)
    for (g, m, v, st, x, B, D) in b:
        eng.lstm_step(spec, wpack, g, m, v, pw, pw, st, x, B, D)
    for sa, sb in zip(a, b):
        for ta, tb in zip(sa[1:5], sb[1:5]):
            assert np.array_equal(eng.to_numpy(ta), eng.to_numpy(tb))


def _run_fused(eng, cfg, params, arrays, x0, B, D, T, state=None, step0=1, Bg=None, x_scale=None):
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D, B_global=Bg, x_scale=x_scale)
    assert eng.unroll_supported(spec, pd)
    if state is None:
        st = eng.state_alloc(B, D)
    else:
        st = eng.state_pack(*[eng.tensor(a) for hc in state for a in hc], B, D)
    x = eng.tensor(x0.reshape(B, D))
    m, v = eng.zeros(B, D), eng.zeros(B, D)
    fx_part = eng.zeros((T + 1) * B)
    eng.unroll(spec, wpack, pd, x, st, m, v, T, step0, fx_part)
    fx = eng.zeros(T + 1)
    eng.reduce_fx(fx_part, T + 1, B, pd.B_global, fx)
    return eng.to_numpy(fx), eng.to_numpy(x), _unpack_state(eng, st, B, D), eng.to_numpy(m), eng.to_numpy(v)


@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
@pytest.mark.parametrize("kind,B,D,M", [("quadratic", 4, 10, None), ("quadratic", 3, 32, None),
                                        ("lasso", 4, 10, None), ("lasso", 3, 50, 25),
                                        ("rastrigin", 4, 2, None), ("rastrigin", 3, 20, None),
                                        ("square_cos", 4, 2, None), ("square_cos", 3, 24, None)])
def test_fused_unroll_vs_oracle(eng, name, kind, B, D, M):
    cfg = ORACLE_CFGS[name]
    params = make_params(cfg, seed=6, trained_like=True)
    prob, x0, arrays = make_problem(kind, B, D, seed=7, M=M)
    T = 20
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, step0=1)
    fx, x, st, m, v = _run_fused(eng, cfg, params, arrays, x0, B, D, T)
    e_fx = rel_err(fx, res.fx)
    e_x = max_abs(x, res.x.reshape(B, D))
    print("fused %s/%s B=%d D=%d: rel fx=%.3g |dx|=%.3g fx0=%.4g fxT=%.4g"
          % (name, kind, B, D, e_fx, e_x, res.fx[0], res.fx[-1]))
    assert e_fx < 1e-5
    assert e_x < 1e-5 * max(1.0, float(np.abs(res.x).max()))
    for l in range(2):
        for i in range(2):
            assert max_abs(st[l][i], res.state[l][i]) < 1e-5 * max(1.0, float(np.abs(res.state[l][i]).max()))
    if cfg.kind == "rnnprop":
        assert max_abs(m, res.m.reshape(B, D)) < 1e-6 * max(1.0, np.abs(res.m).max())


@pytest.mark.parametrize("kind,B,D,M", [("lasso", 5, 40, 24), ("quadratic", 6, 32, None), ("lasso", 3, 300, 100)])
def test_shared_matrix_equals_replicated(eng, kind, B, D, M):
    """L2O_PROB_W_SHARED (one [M, D] matrix, batch stride 0) == the same matrix replicated per
    problem, bit for bit, in l2o_problem_fg and (when the size fits) in the fused unroll."""
    cfg = ORACLE_CFGS["rnnprop"]
    params = make_params(cfg, seed=81, trained_like=True)
    prob, x0, arrays = make_problem(kind, B, D, seed=82, M=M)
    W0 = np.ascontiguousarray(arrays["W"][0])
    rep = dict(arrays, W=np.broadcast_to(W0, arrays["W"].shape).copy())
    sh = dict(arrays, W=W0, w_shared=True)
    outs = []
    for a in (rep, sh):
        pd = device_problem(eng, a, B, D)
        x = eng.tensor(x0.reshape(B, D))
        f, g = eng.zeros(B), eng.zeros(B, D)
        eng.problem_fg(pd, x, f, g)
        res = [eng.to_numpy(f), eng.to_numpy(g)]
        if eng.unroll_supported(spec_of(cfg), pd):
            res += list(_run_fused(eng, cfg, params, a, x0, B, D, 6)[:2])
        outs.append(res)
    assert len(outs[0]) == len(outs[1])
    # l2o_problem_fg reads a per-problem matrix ONCE (k_problem_fg1, D % 4 == 0 and D >= 64) but an L2-resident
    # shared one in two passes (faster there): same numbers up to the summation order; everything else bit for bit
    one_pass = D % 4 == 0 and D >= 64
    for k, (r, s_) in enumerate(zip(*outs)):
        if one_pass and k < 2:
            np.testing.assert_allclose(s_, r, rtol=2e-6, atol=2e-6 * float(np.abs(r).max()))
        else:
            assert np.array_equal(r, s_)


def test_fused_unroll_random_shapes(eng):
    """Seeded sweep over odd shapes (tile counts, ragged last tiles, M != D, batch not a multiple
    of the 8-problem launch groups): fused kernel (pair and single-CU forms) == oracle."""
    import os
    rng = np.random.default_rng(2024)
    cases = []
    for _ in range(14):
        kind = ["quadratic", "lasso", "rastrigin", "square_cos"][int(rng.integers(4))]
        D = int(rng.integers(1, 129))
        B = int(rng.integers(1, 20))
        M = int(rng.integers(1, 16 * ((D + 15) // 16) + 1)) if kind == "lasso" else None
        name = ["dm", "dm_logsign", "rnnprop"][int(rng.integers(3))]
        cases.append((kind, B, D, M, name))
    worst = 0.0
    for kind, B, D, M, name in cases:
        cfg = ORACLE_CFGS[name]
        params = make_params(cfg, seed=B * 131 + D, trained_like=True)
        prob, x0, arrays = make_problem(kind, B, D, seed=D * 7 + B, M=M)
        T = 6
        res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, step0=3)
        for no_pair in (False, True):
            with lib_option(_abi.OPT_PAIR, 0 if no_pair else 1):
                fx, x, st, m, v = _run_fused(eng, cfg, params, arrays, x0, B, D, T, step0=3)
            e = rel_err(fx, res.fx)
            worst = max(worst, e)
            assert e < 1e-5, (kind, B, D, M, name, no_pair, e)
            assert max_abs(x, res.x.reshape(B, D)) < 1e-5 * max(1.0, float(np.abs(res.x).max())), (kind, B, D, M, name)
    print("random-shape sweep: worst rel fx err %.3g over %d cases x 2 kernels" % (worst, len(cases)))


@pytest.mark.parametrize("exact", [0, 1])
@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
def test_wgrad_blocks_equal_the_dense_product(eng, name, exact):
    """l2o_cwlstm_wgrad computes only the tiles of A^T Bm that hold a weight gradient, the rest of G is zero.  With
    L2O_OPT_EXACT_GATES it runs on the fp32 matrix pipe and is bit-equal on those BLOCKS to the dense l2o_atb (same
    arithmetic per tile); by default the products run on the bf16 pipe as a 3-way split (k_atb_bx3): both are compared
    with a float64 product in units of sum |a||b| (what an fp32 dot product's error scales with), ragged row count,
    values of mixed magnitude and a common-sign bias column included; two runs are bit-identical."""
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    fc = cfg.kind == "rnnprop"
    P = 20 if fc else (2 if name == "dm_logsign" else 1)
    H = 20
    K1 = P + H
    KA, KB = K1 + 3 * H + (2 if fc else 0) + 1, 8 * H + 1 + (H if fc else 0)
    rng = np.random.default_rng(5)
    R = 16384 * 3 + 5
    An = np.tanh(rng.standard_normal((R, KA))).astype(np.float32)
    An[:, -1] = 1.0
    Bn = (1e-3 * np.exp(2.0 * rng.standard_normal((R, 1))) * rng.standard_normal((R, KB))).astype(np.float32)
    A, B = eng.tensor(An), eng.tensor(Bn)
    dense = eng.to_numpy(eng.atb(A, B))
    with lib_option(_abi.OPT_EXACT_GATES, exact):
        got = eng.to_numpy(eng.wgrad(spec, A, B))
        again = eng.to_numpy(eng.wgrad(spec, A, B))
    assert np.array_equal(got, again)
    want = An.astype(np.float64).T @ Bn.astype(np.float64)
    mag = np.abs(An).astype(np.float64).T @ np.abs(Bn).astype(np.float64)
    blocks = [(slice(0, K1), slice(0, 4 * H)), (slice(K1, K1 + 2 * H), slice(4 * H, 8 * H)),
              (slice(K1 + 2 * H, K1 + 3 * H), slice(8 * H, 8 * H + 1)), (slice(KA - 1, KA), slice(0, KB))]
    if fc:
        blocks.append((slice(K1 + 3 * H, K1 + 3 * H + 2), slice(8 * H + 1, 8 * H + 1 + H)))
    used = np.zeros((KA, KB), bool)
    worst, worst_dense = 0.0, 0.0
    for r, c in blocks:
        if exact:
            assert np.array_equal(got[r, c], dense[r, c]), (name, r, c)
        worst = max(worst, float((np.abs(got[r, c] - want[r, c]) / mag[r, c]).max()))
        worst_dense = max(worst_dense, float((np.abs(dense[r, c] - want[r, c]) / mag[r, c]).max()))
        used[r, c] = True
    print("%s exact=%d: wgrad err / sum|a||b| %.3g (dense fp32-pipe product %.3g)" % (name, exact, worst, worst_dense))
    assert worst < 1e-6, worst
    tiles = np.zeros((KA, KB), bool)                      # whole 16 x 16 tiles that touch a block are computed
    for i in range(0, KA, 16):
        for j in range(0, KB, 16):
            if used[i:i + 16, j:j + 16].any():
                tiles[i:i + 16, j:j + 16] = True
    assert np.all(got[~tiles] == 0.0)
    assert tiles.sum() < 0.75 * KA * KB


@pytest.mark.parametrize("R", [1, 7, 31, 33, 40, 257])
def test_wgrad_ragged_rows(eng, R):
    """Row counts that end inside a 32-row block / inside a wave's octet of rows (both pipes)."""
    cfg = ORACLE_CFGS["dm"]
    spec = spec_of(cfg)
    KA, KB = 82, 161
    rng = np.random.default_rng(R)
    An = rng.standard_normal((R, KA)).astype(np.float32)
    Bn = rng.standard_normal((R, KB)).astype(np.float32)
    want = An.astype(np.float64).T @ Bn.astype(np.float64)
    for exact in (0, 1):
        with lib_option(_abi.OPT_EXACT_GATES, exact):
            got = eng.to_numpy(eng.wgrad(spec, eng.tensor(An), eng.tensor(Bn)))
        for r, c in [(slice(0, 21), slice(0, 80)), (slice(21, 61), slice(80, 160)), (slice(61, 81), slice(160, 161)),
                     (slice(81, 82), slice(0, 161))]:
            assert float(np.abs(got[r, c] - want[r, c]).max()) < 1e-5 * np.sqrt(R) + 1e-6, (R, exact, r, c)


def test_two_cu_form_big_batches_random(eng):
    """Seeded sweep of the two-CU form where it runs as SEVERAL launches (more than #CU / 2 problems: equal chunks
    of whole launch groups), with x scaling, B_global > B_local and a non-unit step0, against the oracle; the
    chunked launches also agree bit for bit with the same problems run in two separate smaller batches."""
    from open_l2o_amd import _abi
    old_opt = _abi.set_option(_abi.OPT_ONE_LDS, 0)      # (this test is about the CHUNKED two-CU form; k_unroll_lds: test_unroll_lds.py)
    try:
        _two_cu_big_batches(eng)
    finally:
        _abi.set_option(_abi.OPT_ONE_LDS, old_opt)


def _two_cu_big_batches(eng):
    rng = np.random.default_rng(77)
    worst = 0.0
    for case in range(6):
        kind = ["quadratic", "lasso", "square_cos"][case % 3]
        name = ["dm", "dm_logsign", "rnnprop"][int(rng.integers(3))]
        D = int(rng.integers(17, 129))
        B = int(rng.integers(129, 300))
        M = int(rng.integers(4, 16 * ((D + 15) // 16) + 1)) if kind == "lasso" else None
        cfg = ORACLE_CFGS[name]
        params = make_params(cfg, seed=case + 5, trained_like=True)
        prob, x0, arrays = make_problem(kind, B, D, seed=300 + case, M=M)
        prob.batch_global = 2 * B
        xs = np.exp(rng.uniform(-0.3, 0.3, (B, D))).astype(np.float32)
        T = 4
        res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, x_scale=xs, step0=2)
        fx, x, st, m, v = _run_fused(eng, cfg, params, arrays, x0, B, D, T, step0=2, Bg=2 * B, x_scale=xs)
        e = rel_err(fx, res.fx)
        worst = max(worst, e)
        assert e < 1e-5, (kind, name, B, D, M, e)
        assert max_abs(x, res.x.reshape(B, D)) < 1e-5 * max(1.0, float(np.abs(res.x).max())), (kind, name, B, D, M)
        # the first 64 problems alone (ONE launch; same B_global, hence the same scaling): bit-identical iterates
        sub = {k_: (a[:64] if isinstance(a, np.ndarray) and a.shape[:1] == (B,) else a) for k_, a in arrays.items()}
        x_sub = _run_fused(eng, cfg, params, sub, x0.reshape(B, D)[:64], 64, D, T, step0=2, Bg=2 * B, x_scale=xs[:64])[1]
        assert np.array_equal(x_sub, x[:64]), (kind, name, B, D, M)
    print("two-CU form, 129..299 problems (2-3 chunk launches): worst rel fx err %.3g" % worst)


def test_fused_unroll_continuation_and_scale(eng):
    """Two T=10 launches carrying x/state/m/v (the harness' `update`) == one T=20 launch;
    x_scale placeholder chain rule; B_global > B_local."""
    cfg = O.RNNPROP
    params = make_params(cfg, seed=8, trained_like=True)
    B, D = 3, 24
    prob, x0, arrays = make_problem("quadratic", B, D, seed=9)
    prob.batch_global = 2 * B
    rng = np.random.default_rng(10)
    xs = np.exp(rng.uniform(-0.5, 0.5, (B, D))).astype(np.float32)
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), 20, x_scale=xs, step0=1)
    fx, x, st, m, v = _run_fused(eng, cfg, params, arrays, x0, B, D, 20, Bg=2 * B, x_scale=xs)
    assert rel_err(fx, res.fx) < 1e-5
    # continuation
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D, B_global=2 * B, x_scale=xs)
    xd, std = eng.tensor(x0), eng.state_alloc(B, D)
    md, vd = eng.zeros(B, D), eng.zeros(B, D)
    fxs = []
    for seg in range(2):
        fp = eng.zeros(11 * B)
        eng.unroll(spec, wpack, pd, xd, std, md, vd, 10, 1 + 10 * seg, fp)
        f = eng.zeros(11)
        eng.reduce_fx(fp, 11, B, 2 * B, f)
        fxs.append(eng.to_numpy(f))
    np.testing.assert_allclose(fxs[0][:10], fx[:10], rtol=1e-6)
    np.testing.assert_allclose(fxs[0][10], fxs[1][0], rtol=0, atol=0)
    np.testing.assert_allclose(fxs[1], fx[10:], rtol=2e-6)
    np.testing.assert_allclose(eng.to_numpy(xd), x, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("name", ["dm", "rnnprop"])
def test_fused_equals_step_path_at_full_size(eng, name):
    """BASELINE config 2 shape (Quadratic d=128, B=128, T=100): the fused persistent
    kernel and the step-granular kernels (problem_fg + lstm_step) are two independent
    HIP implementations of the same recurrence and must agree."""
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=11, trained_like=True)
    B, D, T = 128, 128, 100
    prob, x0, arrays = make_problem("quadratic", B, D, seed=12)
    fx, x, st, m, v = _run_fused(eng, cfg, params, arrays, x0, B, D, T)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D)
    xd, std, md, vd = eng.tensor(x0), eng.state_alloc(B, D), eng.zeros(B, D), eng.zeros(B, D)
    f, g = eng.zeros(B), eng.zeros(B, D)
    fx2 = eng.zeros(T + 1)
    b95 = float(np.float32(0.95))
    for t in range(T):
        eng.problem_fg(pd, xd, f, g)
        eng.reduce_fx(f, 1, B, B, fx2[t:t + 1])
        eng.lstm_step(spec, wpack, g, md, vd, b95 ** (1 + t), b95 ** (1 + t), std, xd, B, D)
    eng.problem_fg(pd, xd, f, None)
    eng.reduce_fx(f, 1, B, B, fx2[T:T + 1])
    fx2 = eng.to_numpy(fx2)
    e = rel_err(fx, fx2)
    print("fused vs step path %s: rel fx=%.3g, fx0=%.5g fxT=%.5g" % (name, e, fx[0], fx[-1]))
    assert np.all(np.isfinite(fx))
    assert e < 1e-5
    assert max_abs(x, eng.to_numpy(xd)) < 1e-5 * max(1.0, np.abs(x).max())


def test_fused_c2_vs_oracle_trajectory(eng):
    """Config 2 (L2O-DM, Quadratic d=128, B=128, T=100) against the oracle itself."""
    cfg = O.DM_IDENTITY
    params = make_params(cfg, seed=13, trained_like=True)
    B, D, T = 128, 128, 100
    prob, x0, arrays = make_problem("quadratic", B, D, seed=14)
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T)
    fx, x, st, m, v = _run_fused(eng, cfg, params, arrays, x0, B, D, T)
    e = rel_err(fx, res.fx)
    print("C2 fused vs oracle: rel fx=%.3g  fx0=%.5g fxT=%.5g" % (e, res.fx[0], res.fx[-1]))
    assert e < 1e-5


# ---------------------------------------------------------------------------
# BASELINE.json full-size configurations (checked against the multi-threaded C oracle,
# which finishes these sizes in seconds on the GPU box's host)
# ---------------------------------------------------------------------------
def test_c4_rastrigin_shard_full_size(eng):
    """Config 4: L2O-DM on Rastrigin d=100, batch 1024 sharded over 8 GPUs -> this GPU's
    shard is 128 problems with B_global = 1024; T=100."""
    from oracle.c_oracle import c_unroll
    cfg = O.DM_IDENTITY
    params = make_params(cfg, seed=15, trained_like=True)
    B, D, T, Bg = 128, 100, 100, 1024
    prob, x0, arrays = make_problem("rastrigin", B, D, seed=16)
    fx_ref, x_ref, st_ref, _, _, _ = c_unroll("rastrigin", cfg, params, arrays, x0, T, B_global=Bg)
    fx, x, st, m, v = _run_fused(eng, cfg, params, arrays, x0, B, D, T, Bg=Bg)
    e = rel_err(fx, fx_ref)
    print("C4 shard fused vs C oracle: rel fx=%.3g  fx0=%.5g fxT=%.5g" % (e, fx_ref[0], fx_ref[-1]))
    assert np.all(np.isfinite(fx)) and e < 1e-5


def _run_steps(eng, cfg, params, arrays, x0, B, D, T, step0=1, carry=None):
    """Step-granular path (l2o_problem_fg + l2o_cwlstm_step per step)."""
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D)
    if carry is None:
        xd, std, md, vd = eng.tensor(x0.reshape(B, D)), eng.state_alloc(B, D), eng.zeros(B, D), eng.zeros(B, D)
    else:
        xd, std, md, vd = carry
    f, g = eng.zeros(B), eng.zeros(B, D)
    fx = eng.zeros(T + 1)
    b95 = float(np.float32(0.95))
    for t in range(T):
        eng.problem_fg(pd, xd, f, g)
        eng.reduce_fx(f, 1, B, B, fx[t:t + 1])
        eng.lstm_step(spec, wpack, g, md, vd, b95 ** (step0 + t), b95 ** (step0 + t), std, xd, B, D)
    eng.problem_fg(pd, xd, f, None)
    eng.reduce_fx(f, 1, B, B, fx[T:T + 1])
    return eng.to_numpy(fx), (xd, std, md, vd)


_C3_REF = {}


def _c3_reference(cfg, params, arrays, x0):
    """The C oracle's full T = 200 unroll of config 3 (a few seconds on the box's cores), shared by the two
    full-size tests."""
    if "fx" not in _C3_REF:
        from oracle.c_oracle import c_unroll
        fx, x, _, _, _, _ = c_unroll("lasso", cfg, params, arrays, x0, 200)
        _C3_REF.update(fx=fx, x=x)
    return _C3_REF["fx"], _C3_REF["x"]


def test_c3_lasso_rnnprop_full_size(eng):
    """Config 3: L2O-RNNProp on Lasso A in R^{256x512}, lambda=0.1, batch=256, T=200
    (per-problem A: 128 MiB streamed once per step; step-granular kernels).
    The WHOLE T=200 loss trajectory and x_T against the C oracle (1e-5 relative, the north_star
    tolerance), plus the continuation property (200 steps == 2 x 100 steps with carried
    x/state/m/v, bit for bit)."""
    cfg = O.RNNPROP
    params = make_params(cfg, seed=17, trained_like=True)
    B, D, M = 256, 512, 256
    prob, x0, arrays = make_problem("lasso", B, D, seed=18, M=M)
    fx_ref, x_ref = _c3_reference(cfg, params, arrays, x0)
    fx200, carry200 = _run_steps(eng, cfg, params, arrays, x0, B, D, 200)
    e = rel_err(fx200, fx_ref)
    ex = max_abs(eng.to_numpy(carry200[0]), x_ref) / max(1.0, float(np.abs(x_ref).max()))
    print("C3 step path vs C oracle (all 200 steps): rel fx=%.3g |dx_T|=%.3g fx0=%.5g fx200=%.5g"
          % (e, ex, fx_ref[0], fx_ref[-1]))
    assert e < 1e-5 and ex < 1e-5
    fxa, carry = _run_steps(eng, cfg, params, arrays, x0, B, D, 100)
    fxb, _ = _run_steps(eng, cfg, params, arrays, x0, B, D, 100, step0=101, carry=carry)
    np.testing.assert_array_equal(fx200[:101], fxa)
    np.testing.assert_array_equal(fx200[100:], fxb)


@pytest.mark.parametrize("activation,batch", [("sigmoid", 128), ("relu", 100), ("sigmoid", 7)])
@pytest.mark.parametrize("generic", [0, 1])
def test_mlp_fg_kernel(eng, activation, batch, generic):
    """l2o_mlp_fg (problems.mnist forward + gradient) against the oracle: the wave-per-sample kernels for hidden width 20
    and (L2O_OPT_MLP_GENERIC = 1 / l2o_mlp.flags) the generic ones that serve every other shape."""
    from open_l2o_amd._engine import MlpDesc
    from open_l2o_amd import _abi
    with lib_option(_abi.OPT_MLP_GENERIC, generic):
        _mlp_fg_kernel(eng, activation, batch)


def _mlp_fg_kernel(eng, activation, batch):
    from open_l2o_amd._engine import MlpDesc
    rng = np.random.default_rng(80)
    n_data, n_in, H, Oo = 500, 784, 20, 10
    images = rng.random((n_data, n_in)).astype(np.float32)
    labels = rng.integers(0, Oo, n_data).astype(np.int32)
    idx = rng.integers(0, n_data, batch).astype(np.int32)
    ref = O.MnistMLP(images, labels, activation)
    variables = [(rng.standard_normal(s) * 0.3).astype(np.float32) for s in ((n_in, H), (H,), (H, Oo), (Oo,))]
    f_ref, g_ref = ref.fg(variables, idx)
    d = MlpDesc(n_in, H, Oo, batch, 0 if activation == "sigmoid" else 1, eng.tensor(images), eng.int_tensor(labels))
    dv = [eng.tensor(v) for v in variables]
    loss = eng.zeros(1)
    grads = [eng.zeros(*v.shape) for v in variables]
    eng.mlp_fg(d, eng.int_tensor(idx), *dv, loss, grads)
    assert rel_err(eng.to_numpy(loss)[0], f_ref) < 5e-6
    for g, gr in zip(grads, g_ref):
        assert max_abs(eng.to_numpy(g), gr) < 5e-6 * max(1.0, float(np.abs(gr).max()))
    loss2 = eng.zeros(1)
    eng.mlp_fg(d, eng.int_tensor(idx), *dv, loss2, None)            # forward only
    assert eng.to_numpy(loss2)[0] == eng.to_numpy(loss)[0]


@pytest.mark.parametrize("hidden,activation,batch", [((20, 20), "sigmoid", 128), ((20, 20), "relu", 50), ((32, 8, 20), "sigmoid", 7),
                                                     ((20,), "sigmoid", 64)])
def test_mlp_deep_fg_kernel(eng, hidden, activation, batch):
    """l2o_mlp_deep_fg (problems.mnist with several hidden layers: "mnist_deeper", DM/util.py:157-163) against the oracle."""
    from open_l2o_amd._engine import MlpDeepDesc
    rng = np.random.default_rng(84)
    n_data, n_in, Oo = 400, 784, 10
    images = rng.random((n_data, n_in)).astype(np.float32)
    labels = rng.integers(0, Oo, n_data).astype(np.int32)
    idx = rng.integers(0, n_data, batch).astype(np.int32)
    ref = O.MnistMLP(images, labels, activation)
    widths = [n_in] + list(hidden) + [Oo]
    variables = []
    for l in range(len(widths) - 1):
        variables += [(rng.standard_normal((widths[l], widths[l + 1])) * 0.3).astype(np.float32),
                      (rng.standard_normal(widths[l + 1]) * 0.3).astype(np.float32)]
    f_ref, g_ref = ref.fg_deep(variables, idx)
    d = MlpDeepDesc(n_in, tuple(hidden), Oo, batch, 0 if activation == "sigmoid" else 1, eng.tensor(images), eng.int_tensor(labels))
    dv = [eng.tensor(v) for v in variables]
    loss = eng.zeros(1)
    grads = [eng.zeros(*v.shape) for v in variables]
    eng.mlp_deep_fg(d, eng.int_tensor(idx), dv, loss, grads)
    assert rel_err(eng.to_numpy(loss)[0], f_ref) < 5e-6
    for g, gr in zip(grads, g_ref):
        assert max_abs(eng.to_numpy(g), gr) < 5e-6 * max(1.0, float(np.abs(gr).max()))
    loss2 = eng.zeros(1)
    eng.mlp_deep_fg(d, eng.int_tensor(idx), dv, loss2, None)           # forward only
    assert eng.to_numpy(loss2)[0] == eng.to_numpy(loss)[0]


# ---------------------------------------------------------------------------
# edge cases: degenerate and maximum sizes, both fused variants
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("kind,B,D,M,T", [
    ("quadratic", 1, 1, None, 5),        # one problem, one coordinate (the reference's problems_test shape)
    ("quadratic", 1, 16, None, 0),       # T = 0: only f(x_0)
    ("lasso", 2, 17, 3, 4),              # ragged tile (17 = 16 + 1), far fewer rows than columns
    ("rastrigin", 7, 100, None, 3),      # 7 tiles -> padded to 8 in the two-CU variant
    ("quadratic", 300, 48, None, 3),     # more problems than CUs: one workgroup per problem
    ("lasso", 5, 128, 128, 3),           # the largest fused size (8 tiles, 128 x 128 in LDS)
])
@pytest.mark.parametrize("pair", [True, False])
def test_fused_edge_cases(eng, kind, B, D, M, T, pair):
    cfg = O.DM_LOGSIGN
    params = make_params(cfg, seed=90, trained_like=True)
    prob, x0, arrays = make_problem(kind, B, D, seed=91, M=M)
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T)
    with lib_option(_abi.OPT_PAIR, 1 if pair else 0):
        fx, x, st, m, v = _run_fused(eng, cfg, params, arrays, x0, B, D, T)
    assert fx.shape == (T + 1,)
    assert rel_err(fx, res.fx) < 1e-5
    assert max_abs(x, res.x.reshape(B, D)) < 1e-5 * max(1.0, float(np.abs(res.x).max()))


def test_unroll_follows_a_problem_changed_in_place(eng):
    """The fused unroll keeps no per-problem cache (the normal-matrix form that had one -- l2o_unroll_prepare -- was
    removed with ABI v12): a problem changed IN PLACE between two launches gives the NEW problem's trajectory."""
    cfg = O.DM_IDENTITY
    params = make_params(cfg, seed=21, trained_like=True)
    B, D, T = 6, 40, 8
    prob, x0, arrays = make_problem("quadratic", B, D, seed=22)
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D)
    fx = eng.zeros(T + 1)

    def launch():
        x, st = eng.tensor(x0.reshape(B, D)), eng.state_alloc(B, D)
        eng.unroll(spec, wpack, pd, x, st, None, None, T, 1, eng.zeros((T + 1) * B), fx=fx)
        return eng.to_numpy(fx)

    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T)
    assert rel_err(launch(), res.fx) < 1e-5
    pd.W.mul_(1.25); pd.y.add_(0.5)                                                # the problem changes IN PLACE
    prob2 = O.Quadratic(prob.w * np.float32(1.25), prob.y + np.float32(0.5))
    res2 = O.unroll(prob2, cfg, params, x0, O.net_initial_state(cfg, B * D), T)
    assert rel_err(res2.fx, res.fx) > 1e-2
    assert rel_err(launch(), res2.fx) < 1e-5


def test_fused_rejects_what_it_cannot_do(eng):
    from open_l2o_amd import _abi
    cfg = O.DM_IDENTITY
    spec = spec_of(cfg)
    for kind, B, D, M in (("quadratic", 2, 129, None), ("lasso", 2, 18, 40), ("quadratic", 1, 516, None)):
        prob, x0, arrays = make_problem(kind, B, D, seed=92, M=M)
        pd = device_problem(eng, arrays, B, D)
        assert not eng.unroll_supported(spec, pd)
        with pytest.raises(_abi.L2OUnsupported):
            eng.unroll(spec, eng.pack_weights(spec, make_params(cfg, 1)), pd, eng.tensor(x0.reshape(B, D)),
                       eng.state_alloc(B, D), None, None, 2, 1, eng.zeros(3 * B))
        # ... and the step-granular kernels take over
        fx, _ = _run_steps(eng, cfg, make_params(cfg, 1, trained_like=True), arrays, x0, B, D, 3)
        res = O.unroll(prob, cfg, make_params(cfg, 1, trained_like=True), x0, O.net_initial_state(cfg, B * D), 3)
        assert rel_err(fx, res.fx) < 1e-5


# ---------------------------------------------------------------------------
# the streaming form of the fused unroll (csrc/l2o_unroll_cu.h): 128 < D <= 512, any M
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,kind,B,D,M,T", [
    ("rnnprop", "lasso", 3, 512, 256, 20),        # BASELINE config 3's problem shape: 32 tiles, the 8th per wave in registers
    ("dm", "quadratic", 2, 256, None, 20),        # one float4 chunk per lane (NV = 1)
    ("dm_logsign", "rastrigin", 3, 200, None, 12),    # ragged last tile (200 = 12 x 16 + 8), tiles not a multiple of 4
    ("rnnprop", "square_cos", 2, 132, None, 12),  # the smallest sizes: 9 tiles
    ("dm", "lasso", 2, 452, 37, 8),               # odd row count (not a multiple of the 4-row groups), 29 tiles
    ("dm_logsign", "lasso", 2, 480, 500, 8),      # more rows than columns, 30 tiles (8th slot on waves 0, 1 only)
    ("rnnprop", "quadratic", 1, 512, None, 0),    # T = 0: only f(x_0)
    ("dm", "lasso", 2, 16, 40, 6),                # D <= 128 with more rows than the LDS-resident forms take
    ("rnnprop", "lasso", 3, 64, 200, 6),
])
def test_streaming_unroll_vs_oracle(eng, name, kind, B, D, M, T):
    cfg = ORACLE_CFGS[name]
    params = make_params(cfg, seed=31, trained_like=True)
    prob, x0, arrays = make_problem(kind, B, D, seed=32, M=M)
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, step0=1)
    fx, x, st, m, v = _run_fused(eng, cfg, params, arrays, x0, B, D, T)
    e_fx = rel_err(fx, res.fx)
    e_x = max_abs(x, res.x.reshape(B, D))
    print("streaming %s/%s B=%d D=%d: rel fx=%.3g |dx|=%.3g fx0=%.4g fxT=%.4g"
          % (name, kind, B, D, e_fx, e_x, res.fx[0], res.fx[-1]))
    assert e_fx < 1e-5
    assert e_x < 1e-5 * max(1.0, float(np.abs(res.x).max()))
    for l in range(2):
        for i in range(2):
            assert max_abs(st[l][i], res.state[l][i]) < 1e-5 * max(1.0, float(np.abs(res.state[l][i]).max()))
    if cfg.kind == "rnnprop":
        assert max_abs(m, res.m.reshape(B, D)) < 1e-6 * max(1.0, np.abs(res.m).max())
        assert max_abs(v, res.v.reshape(B, D)) < 1e-6 * max(1.0, np.abs(res.v).max())


def test_streaming_unroll_continuation_and_step_path(eng):
    """Streaming form: 16 steps == 2 x 8 steps with carried x / state / m / v (bit for bit: nothing
    in the kernel depends on T), with a random x-scale and a batch that is a shard of a larger one;
    and == the step-granular path (l2o_problem_fg + l2o_cwlstm_step) to fp32 summation-order accuracy."""
    cfg = O.RNNPROP
    params = make_params(cfg, seed=41, trained_like=True)
    B, D, M = 5, 384, 96
    prob, x0, arrays = make_problem("lasso", B, D, seed=42, M=M)
    xs = np.exp(np.random.default_rng(43).uniform(-1, 1, (B, D))).astype(np.float32)
    fx16, x16, st16, m16, v16 = _run_fused(eng, cfg, params, arrays, x0, B, D, 16, Bg=4 * B, x_scale=xs)
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D, B_global=4 * B, x_scale=xs)
    st = eng.state_alloc(B, D)
    x = eng.tensor(x0.reshape(B, D))
    m, v = eng.zeros(B, D), eng.zeros(B, D)
    fxs = []
    for k in range(2):
        fx_part, fx = eng.zeros(9 * B), eng.zeros(9)
        eng.unroll(spec, wpack, pd, x, st, m, v, 8, 1 + 8 * k, fx_part)
        eng.reduce_fx(fx_part, 9, B, pd.B_global, fx)
        fxs.append(eng.to_numpy(fx))
    # (not bit for bit: a launch starts beta^step0 from pow() and carries it as a running float-float product)
    fx88 = np.concatenate([fxs[0], fxs[1][1:]])
    assert fxs[0][8] == fxs[1][0] and rel_err(fx88, fx16) < 1e-6
    for got, ref in ((eng.to_numpy(x), x16), (eng.to_numpy(m), m16), (eng.to_numpy(v), v16)):
        assert max_abs(got, ref) < 1e-6 * max(1.0, float(np.abs(ref).max()))
    # the step-granular path on the same (unscaled, unsharded) inputs
    fxf, xf = _run_fused(eng, cfg, params, arrays, x0, B, D, 16)[:2]
    fxs_, carry = _run_steps(eng, cfg, params, arrays, x0, B, D, 16)
    assert rel_err(fxf, fxs_) < 1e-5
    xs_ = eng.to_numpy(carry[0])
    assert max_abs(xf, xs_) < 1e-5 * max(1.0, float(np.abs(xs_).max()))


def test_c3_streaming_full_size(eng):
    """Config 3 at full size (RNNProp, Lasso 256 x 512 per problem, batch 256, T = 200) through the
    streaming fused unroll: the WHOLE T = 200 loss trajectory and x_T against the C oracle, and a
    20-step launch equal to the head of the 200-step one bit for bit."""
    cfg = O.RNNPROP
    params = make_params(cfg, seed=17, trained_like=True)
    B, D, M = 256, 512, 256
    prob, x0, arrays = make_problem("lasso", B, D, seed=18, M=M)
    fx_ref, x_ref = _c3_reference(cfg, params, arrays, x0)
    fx200, x200 = _run_fused(eng, cfg, params, arrays, x0, B, D, 200)[:2]
    e = rel_err(fx200, fx_ref)
    ex = max_abs(x200, x_ref) / max(1.0, float(np.abs(x_ref).max()))
    print("C3 streaming unroll vs C oracle (all 200 steps): rel fx=%.3g |dx_T|=%.3g fx0=%.5g fx200=%.5g"
          % (e, ex, fx_ref[0], fx_ref[-1]))
    assert e < 1e-5 and ex < 1e-5
    fx20 = _run_fused(eng, cfg, params, arrays, x0, B, D, 20)[0]
    np.testing.assert_array_equal(fx200[:21], fx20)


@pytest.mark.parametrize("name,kind,B,D,M", [("dm", "quadratic", 2, 256, None), ("rnnprop", "lasso", 3, 512, 40),
                                             ("dm_logsign", "lasso", 2, 200, 24),
                                             ("rnnprop", "quadratic", 150, 40, None),   # two-CU form, 2 chunk launches
                                             ("dm", "quadratic", 131, 20, None)])
def test_streaming_unroll_records_history(eng, name, kind, B, D, M):
    """l2o_unroll_record on the streaming form (D > 128) and on the two-CU form with more problems than one launch
    holds (consecutive chunk launches): the recorded history -- packed state BEFORE each step,
    the gradient fed to the network, RNNProp moments AFTER the step, the gradient at x_T -- equals what the
    step-granular kernels see along the same trajectory, and the recording launch leaves the same x / fx as
    the plain one."""
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=61, trained_like=True)
    prob, x0, arrays = make_problem(kind, B, D, seed=62, M=M)
    pd = device_problem(eng, arrays, B, D)
    assert eng.unroll_supported(spec, pd) and eng.unroll_supported(spec, pd, record=True)
    T, N = 5, B * D
    wpack = eng.pack_weights(spec, params)
    x, st = eng.tensor(x0.reshape(B, D)), eng.state_alloc(B, D)
    m, v = eng.zeros(B, D), eng.zeros(B, D)
    fx_part = eng.zeros((T + 1) * B)
    hist = {"st": eng.zeros(T, st.numel()), "g": eng.zeros(T, N), "g_final": eng.zeros(N)}
    if name == "rnnprop":
        hist.update(m=eng.zeros(T, N), v=eng.zeros(T, N))
    eng.unroll(spec, wpack, pd, x, st, m, v, T, 2, fx_part, hist=hist)
    fx_plain, x_plain = _run_fused(eng, cfg, params, arrays, x0, B, D, T, step0=2)[:2]
    # (the recording variant is a different instantiation: same arithmetic, the compiler may contract differently)
    np.testing.assert_allclose(eng.to_numpy(x), x_plain, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(eng.to_numpy(fx_part).reshape(T + 1, B).sum(1) / np.float32(B), fx_plain, rtol=1e-6)
    # replay with the step-granular kernels
    xd, std, md, vd = eng.tensor(x0.reshape(B, D)), eng.state_alloc(B, D), eng.zeros(B, D), eng.zeros(B, D)
    f, g = eng.zeros(B), eng.zeros(B, D)
    b95 = float(np.float32(0.95))
    for t in range(T):
        eng.problem_fg(pd, xd, f, g)
        assert max_abs(eng.to_numpy(hist["st"][t]), eng.to_numpy(std)) < 2e-6, t
        gs = float(np.abs(eng.to_numpy(g)).max())
        assert max_abs(eng.to_numpy(hist["g"][t]), eng.to_numpy(g).reshape(-1)) < 2e-5 * gs, t
        eng.lstm_step(spec, wpack, g, md, vd, b95 ** (2 + t), b95 ** (2 + t), std, xd, B, D)
        if name == "rnnprop":
            assert max_abs(eng.to_numpy(hist["m"][t]), eng.to_numpy(md).reshape(-1)) < 2e-5 * gs
            assert max_abs(eng.to_numpy(hist["v"][t]), eng.to_numpy(vd).reshape(-1)) < 4e-5 * gs * gs
    eng.problem_fg(pd, xd, f, g)
    assert max_abs(eng.to_numpy(hist["g_final"]), eng.to_numpy(g).reshape(-1)) < 2e-5 * float(np.abs(eng.to_numpy(g)).max())


@pytest.mark.parametrize("B,T", [(16, 1000), (4, 10000)])
def test_long_horizon(eng, B, T):
    """The curriculum's longest training horizon (DM/train_dm.py:66: num_unrolls up to 50 x unroll_length 20 =
    1000 steps) and the evaluation drivers' 10 000 (DM/evaluate_dm.py:43), each in ONE launch of the fused
    kernels: L2O-DM on Quadratic d = 128 (config-2 shape).  fp32 trajectories drift from the exact one; the
    bound is the drift of the fp32 ORACLE from its own float64 evaluation (x 3), never looser than 1e-4
    (measured 6e-7 for the oracle at T = 1000); the first 101 steps hold the 1e-5 of the short tests."""
    from oracle.c_oracle import c_unroll
    cfg = O.DM_IDENTITY
    params = make_params(cfg, seed=3, trained_like=True)
    D = 128
    prob, x0, arrays = make_problem("quadratic", B, D, seed=4)
    fx32 = c_unroll("quadratic", cfg, params, arrays, x0, T)[0]
    p64 = {k: {v: a.astype(np.float64) for v, a in d.items()} for k, d in params.items()}
    r64 = O.unroll(O.Quadratic(prob.w.astype(np.float64), prob.y.astype(np.float64)), cfg, p64, x0.astype(np.float64),
                   O.net_initial_state(cfg, B * D, np.float64), T)
    envelope = rel_err(fx32, r64.fx)
    for pair in (1, 0):
        with lib_option(_abi.OPT_PAIR, pair):
            fx = _run_fused(eng, cfg, params, arrays, x0, B, D, T)[0]
        e64, e32 = rel_err(fx, r64.fx), rel_err(fx, fx32)
        print("T=%d (pair=%d): rel fx vs float64 oracle %.3g, vs fp32 C oracle %.3g (fp32 oracle's own drift %.3g)"
              % (T, pair, e64, e32, envelope))
        assert e64 < min(1e-4, max(1e-5, 3 * envelope))
        assert rel_err(fx[:101], r64.fx[:101]) < 1e-5


@pytest.mark.parametrize("R,KA,KB", [(16384 * 5, 82, 161), (4000, 83, 161), (16384 * 3 + 7, 103, 181), (50, 2, 1),
                                     (31, 82, 161), (100000, 21, 80)])
def test_atb_weight_gradient_contraction(eng, R, KA, KB):
    """l2o_atb (split-K A^T B on the fp32 matrix cores) against a float64 product, for the three (KA, KB) of the
    BPTT rows, ragged row counts and the tiny Linear-only case; two runs are bit-identical (fixed-order reduction)."""
    import torch
    rng = np.random.default_rng(R + KA)
    A = rng.standard_normal((R, KA)).astype(np.float32)
    B = rng.standard_normal((R, KB)).astype(np.float32)
    Ad, Bd = eng.tensor(A), eng.tensor(B)
    got = eng.to_numpy(eng.atb(Ad, Bd))
    again = eng.to_numpy(eng.atb(Ad, Bd))
    want = A.astype(np.float64).T @ B.astype(np.float64)
    scale = np.sqrt(R)
    assert got.shape == (KA, KB) and np.array_equal(got, again)
    assert float(np.abs(got - want).max()) < 2e-5 * scale, float(np.abs(got - want).max())


@pytest.mark.parametrize("name,kind,B,D,M", [("dm", "quadratic", 128, 128, None), ("rnnprop", "rastrigin", 128, 100, None),
                                             ("dm_logsign", "lasso", 200, 48, None), ("rnnprop", "lasso", 4, 384, 40)])
def test_unroll_reduce_restart_equals_rewind_then_unroll(eng, name, kind, B, D, M):
    """l2o_unroll_reduce(x0, L2O_UNROLL_ZERO_STATE): `reset` of the iterate and of the LSTM state / moments folded into
    the unroll (no copy / memset pass) == an unroll from buffers that hold x0 and zeros, bit for bit -- for the
    two-CU form, the one-CU form (B > #CUs / 2) and the streaming form; fx == l2o_reduce_fx of fx_part."""
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=41, trained_like=True)
    prob, x0, arrays = make_problem(kind, B, D, seed=42, M=M)
    pd = device_problem(eng, arrays, B, D)
    wpack = eng.pack_weights(spec, params)
    T = 7
    x0d = eng.tensor(x0.reshape(B, D))
    # reference: clean buffers, separate reduction
    x, st, m, v = x0d.clone(), eng.state_alloc(B, D), eng.zeros(B, D), eng.zeros(B, D)
    fx_part, fx = eng.zeros((T + 1) * B), eng.zeros(T + 1)
    eng.unroll(spec, wpack, pd, x, st, m, v, T, 3, fx_part)
    eng.reduce_fx(fx_part, T + 1, B, B, fx)
    # restart form: the in-out buffers start with garbage
    g = torch.Generator(device="cpu").manual_seed(1)
    junk = lambda t: torch.randn(t.shape, generator=g).to(t.device)
    x2, st2, m2, v2 = junk(x), junk(st), junk(m).abs(), junk(v).abs()
    fx_part2, fx2 = eng.zeros((T + 1) * B), eng.zeros(T + 1)
    eng.unroll(spec, wpack, pd, x2, st2, m2, v2, T, 3, fx_part2, fx=fx2, x0=x0d, zero_state=True)
    for a, b in ((x, x2), (st, st2), (fx_part, fx_part2), (fx, fx2)) + (((m, m2), (v, v2)) if name == "rnnprop" else ()):
        assert torch.equal(a, b)
    assert torch.equal(x0d, eng.tensor(x0.reshape(B, D)))                # x0 is read-only


def test_guarded_adam_skips_the_update_of_a_failed_unroll(eng):
    """l2o_adam_step_guarded: the update runs iff the status word at the head of the unroll workspace is zero -- what
    lets meta_minimize enqueue the meta-step behind the unroll without waiting for the status on the host."""
    rng = np.random.default_rng(9)
    n = 1000
    w0, g = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    ws = torch.zeros(64, dtype=torch.uint8, device=eng.device)
    old = eng._last_ws
    eng._last_ws = ws
    try:
        w, m, v = eng.tensor(w0), eng.zeros(n), eng.zeros(n)
        ws[:4] = torch.tensor([1, 0, 0, 0], dtype=torch.uint8)          # "partner timeout"
        eng.adam_step(w, m, v, eng.tensor(g), 0.01, 0.9, 0.999, 1e-8, guarded=True)
        assert np.array_equal(eng.to_numpy(w), w0) and not eng.to_numpy(m).any() and not eng.to_numpy(v).any()
        ws[:4] = 0
        eng.adam_step(w, m, v, eng.tensor(g), 0.01, 0.9, 0.999, 1e-8, guarded=True)
        w2, m2, v2 = eng.tensor(w0), eng.zeros(n), eng.zeros(n)
        eng.adam_step(w2, m2, v2, eng.tensor(g), 0.01, 0.9, 0.999, 1e-8)
        assert np.array_equal(eng.to_numpy(w), eng.to_numpy(w2)) and not np.array_equal(eng.to_numpy(w), w0)
        assert np.array_equal(eng.to_numpy(m), eng.to_numpy(m2)) and np.array_equal(eng.to_numpy(v), eng.to_numpy(v2))
    finally:
        eng._last_ws = old
