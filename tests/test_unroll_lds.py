"""The fused unroll for LARGE shards (round 4; 5..8 tiles; DM nets and RNNProp): k_unroll_lds -- one problem per CU, the
bf16x3 gate GEMM's fragments in LDS, two waves of the same problem per SIMD (L2O_OPT_ONE_LDS = 2: always; the default 1:
above #CU / 2 problems).  It is also the exchange-free form the host falls back to after a partner timeout of the two-CU
kernel (tests/test_recovery.py).  (Round 4's second form, k_unroll_pair2, was removed in round 5.)
Parity against the oracle for every optimizee and both DM preprocessings, ragged sizes, x scaling / B_global /
continuation, the recording form, and against the chunked two-CU form (0)."""
import numpy as np
import pytest

import oracle as O
from helpers import ORACLE_CFGS, device_problem, lib_option, make_params, make_problem, max_abs, rel_err, spec_of
from open_l2o_amd import _abi
from test_hip_kernels import _run_fused

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from open_l2o_amd._engine import HipEngine
    return HipEngine()


@pytest.mark.parametrize("form", [2])
@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
@pytest.mark.parametrize("kind,B,D,M", [("quadratic", 5, 128, None), ("quadratic", 3, 65, None), ("lasso", 4, 100, 70),
                                        ("lasso", 3, 128, 128), ("rastrigin", 6, 100, None), ("square_cos", 3, 81, None),
                                        ("rastrigin", 2, 113, None)])
def test_forced_lds_form_vs_oracle(eng, name, kind, B, D, M, form):
    cfg = ORACLE_CFGS[name]
    params = make_params(cfg, seed=16, trained_like=True)
    prob, x0, arrays = make_problem(kind, B, D, seed=17, M=M)
    T = 20
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, step0=1)
    with lib_option(_abi.OPT_ONE_LDS, form):
        fx, x, st, m, v = _run_fused(eng, cfg, params, arrays, x0, B, D, T)
    e_fx, e_x = rel_err(fx, res.fx), max_abs(x, res.x.reshape(B, D))
    if cfg.kind == "rnnprop":
        assert max_abs(m, res.m.reshape(B, D)) < 2e-6 * max(1.0, float(np.abs(res.m).max()))
        assert max_abs(v, res.v.reshape(B, D)) < 4e-6 * max(1.0, float(np.abs(res.v).max()))
    print("k_unroll_lds %s/%s B=%d D=%d: rel fx=%.3g |dx|=%.3g" % (name, kind, B, D, e_fx, e_x))
    assert e_fx < 1e-5
    assert e_x < 1e-5 * max(1.0, float(np.abs(res.x).max()))
    # (the cos optimizees turn an fp32-rounding difference of x into a larger one of the LSTM state: alpha = 10 times the
    #  curvature 4 pi^2 of the cos term; the losses and iterates above hold 1e-5, the states 1e-5 / 5e-5)
    st_tol = 5e-5 if kind in ("rastrigin", "square_cos") else 1e-5
    if cfg.kind == "rnnprop" and st_tol > 1e-5:
        st_tol = 2e-4     # (RNNProp's inputs g / (sqrt(v^) + eps) amplify it further; the two-CU kernel measures 5.4e-5 here too)
    for l in range(2):
        for i in range(2):
            assert max_abs(st[l][i], res.state[l][i]) < st_tol * max(1.0, float(np.abs(res.state[l][i]).max()))


@pytest.mark.parametrize("name", ["dm", "rnnprop"])
def test_large_shard_forms_equal_chunked_two_cu_form_and_oracle(eng, name):
    """A shard of 300 problems (more than #CU / 2): the default (1) runs k_unroll_lds, 0 the chunked two-CU form; each
    against the oracle (x scaling, B_global > B_local, step0); k_unroll_lds sums the GEMVs in another order than the
    two-CU kernel; continuation (2 x T/2 == T) bit for bit."""
    cfg = ORACLE_CFGS[name]
    params = make_params(cfg, seed=21, trained_like=True)
    B, D, T = 300, 128, 10
    prob, x0, arrays = make_problem("quadratic", B, D, seed=22)
    prob.batch_global = 2 * B
    xs = np.exp(np.random.default_rng(3).uniform(-0.3, 0.3, (B, D))).astype(np.float32)
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, x_scale=xs, step0=1)
    out = {}
    for mode in (1, 0):
        with lib_option(_abi.OPT_ONE_LDS, mode):
            out[mode] = _run_fused(eng, cfg, params, arrays, x0, B, D, T, Bg=2 * B, x_scale=xs)
        assert rel_err(out[mode][0], res.fx) < 1e-5, mode
        assert max_abs(out[mode][1], res.x.reshape(B, D)) < 1e-5 * max(1.0, float(np.abs(res.x).max()))
    print("k_unroll_lds vs chunked two-CU: rel fx %.3g" % rel_err(out[1][0], out[0][0]))
    assert rel_err(out[1][0], out[0][0]) < 2e-6
    # continuation: two launches of T / 2 carrying x and the LSTM state == one launch of T
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D, B_global=2 * B, x_scale=xs)
    for mode in (1,):
        with lib_option(_abi.OPT_ONE_LDS, mode):
            x, st = eng.tensor(x0.reshape(B, D)), eng.state_alloc(B, D)
            m, v = eng.zeros(B, D), eng.zeros(B, D)
            fxp = eng.zeros((T // 2 + 1) * B)
            for k in range(2):                                  # (RNNProp's bias corrections continue at step 1 + T / 2)
                eng.unroll(spec, wpack, pd, x, st, m, v, T // 2, 1 + k * (T // 2), fxp)
            if cfg.kind == "rnnprop":     # (the second launch's beta^step0 comes from the host, the one launch carries the product)
                np.testing.assert_allclose(eng.to_numpy(x), out[mode][1], rtol=2e-6, atol=1e-7)
            else:
                assert np.array_equal(eng.to_numpy(x), out[mode][1]), mode
    eng.check_unroll_status()


@pytest.mark.parametrize("name,form", [("dm_logsign", 1), ("rnnprop", 1)])
def test_recording_form_equals_plain_unroll_prefixes(eng, name, form):
    """HIST instantiation (l2o_unroll_record on a large shard): the recording launch leaves the same x / fx as the plain
    one (a different instantiation: same arithmetic, the compiler may contract differently -> 1e-5), the recorded state
    BEFORE step t is the state a t-step plain unroll ends with (RNNProp: the recorded moments AFTER step t - 1 are its
    moments), the recorded gradients are the optimizee's gradients at the recorded iterates."""
    cfg = ORACLE_CFGS[name]
    rp = cfg.kind == "rnnprop"
    params = make_params(cfg, seed=31, trained_like=True)
    B, D, T = 160, 100, 6
    prob, x0, arrays = make_problem("rastrigin", B, D, seed=32)
    spec = spec_of(cfg)
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D)
    assert eng.unroll_supported(spec, pd, record=True)

    def run(t, hist=None):
        x, st = eng.tensor(x0.reshape(B, D)), eng.state_alloc(B, D)
        m, v = eng.zeros(B, D), eng.zeros(B, D)
        fxp = eng.zeros((t + 1) * B)
        with lib_option(_abi.OPT_ONE_LDS, form):
            eng.unroll(spec, wpack, pd, x, st, m, v, t, 1, fxp, hist=hist)
        return eng.to_numpy(x), eng.to_numpy(fxp), eng.to_numpy(st), eng.to_numpy(m), eng.to_numpy(v)

    hist = dict(st=eng.zeros(T, eng.state_floats(B, D)), g=eng.zeros(T, B * D), g_final=eng.zeros(B * D))
    if rp:
        hist.update(m=eng.zeros(T, B * D), v=eng.zeros(T, B * D))
    x_rec, fx_rec = run(T, hist)[:2]
    x_pl, fx_pl = run(T)[:2]
    np.testing.assert_allclose(x_rec, x_pl, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(fx_rec, fx_pl, rtol=1e-5)
    for t in (1, 3, 5):
        x_t, _, st_t, m_t, v_t = run(t)
        assert max_abs(eng.to_numpy(hist["st"][t]), st_t) < 3e-6, t
        g_t = prob.grad(x_t.reshape(x0.shape)).reshape(-1)
        gs = float(np.abs(g_t).max())
        assert max_abs(eng.to_numpy(hist["g"][t]), g_t) < 2e-5 * gs, t
        if rp:
            assert max_abs(eng.to_numpy(hist["m"][t - 1]), m_t.reshape(-1)) < 2e-5 * gs, t
            assert max_abs(eng.to_numpy(hist["v"][t - 1]), v_t.reshape(-1)) < 4e-5 * gs * gs, t
    g0 = prob.grad(x0).reshape(-1)
    assert max_abs(eng.to_numpy(hist["g"][0]), g0) < 2e-6 * float(np.abs(g0).max())
    gT = prob.grad(x_pl.reshape(x0.shape)).reshape(-1)
    assert max_abs(eng.to_numpy(hist["g_final"]), gT) < 2e-5 * float(np.abs(gT).max())
    eng.check_unroll_status()


@pytest.mark.parametrize("form", [2])
def test_shared_matrix_equals_replicated(eng, form):
    """L2O_PROB_W_SHARED (one [M, D] matrix for the whole batch, DM/problems.py lasso_fixed) through the LDS-fragment
    kernels == the same matrix replicated per problem, bit for bit."""
    cfg = ORACLE_CFGS["dm_logsign"]
    params = make_params(cfg, seed=41, trained_like=True)
    B, D, M, T = 5, 100, 70, 8
    prob, x0, arrays = make_problem("lasso", B, D, seed=42, M=M)
    W0 = np.ascontiguousarray(arrays["W"][0])
    rep = dict(arrays, W=np.broadcast_to(W0, arrays["W"].shape).copy())
    sh = dict(arrays, W=W0, w_shared=True)
    with lib_option(_abi.OPT_ONE_LDS, form):
        a = _run_fused(eng, cfg, params, rep, x0, B, D, T)
        b = _run_fused(eng, cfg, params, sh, x0, B, D, T)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    prob.w[:] = W0
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, step0=1)
    assert rel_err(a[0], res.fx) < 1e-5


def test_large_shard_through_the_product_api(eng):
    """MetaOptimizer.meta_loss on 300 Rastrigin problems of d = 100 (config 4's shape, a shard larger than #CU / 2): the
    default path is k_unroll_lds; Session.run([loss, fx, x, update]) twice (continuation) and UnrollGraph.launch(restart=)
    -- the prepared-call path bench.py uses, with the zero-state / x0 rewind folded into the launch -- against the oracle."""
    from open_l2o_amd import _engine, meta, problems
    from open_l2o_amd.session import Session
    from test_meta_api import _net_config
    old = _engine._default_engine
    _engine.set_default_engine(eng)
    try:
        cfg = ORACLE_CFGS["dm"]
        params = make_params(cfg, seed=51, trained_like=True)
        B, D, T = 300, 100, 10
        prob, x0, _ = make_problem("rastrigin", B, D, seed=52)
        problem = problems.rastrigin(batch_size=B, num_dims=D, data={"A": prob.A, "B": prob.B, "C": prob.C, "x": x0})
        optimizer = meta.MetaOptimizer(**_net_config(cfg, params))
        ml = optimizer.meta_loss(problem, T)
        res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T)
        res2 = O.unroll(prob, cfg, params, res.x, res.state, T)
        with Session() as sess:
            sess.run(ml.reset)
            l1, f1, x1, _ = sess.run([ml.loss, ml.fx, ml.x, ml.update])
            l2, f2, _, _ = sess.run([ml.loss, ml.fx, ml.x, ml.update])
        g = optimizer.graph
        assert g.last_path == "fused"
        assert rel_err(l1, res.loss) < 1e-5 and rel_err(f1, res.fx[-1]) < 1e-5
        assert rel_err(l2, res2.loss) < 2e-5 and rel_err(f2, res2.fx[-1]) < 2e-5
        np.testing.assert_allclose(x1[0], res.x, rtol=1e-4, atol=2e-6)
        x0d = [eng.tensor(x0)]
        outs = []
        for _ in range(3):
            fx, xs = g.launch({}, commit=True, restart=x0d)
            outs.append((eng.to_numpy(fx).copy(), eng.to_numpy(xs[0]).copy()))
        assert rel_err(outs[0][0], res.fx) < 1e-5
        for fxk, xk in outs[1:]:
            assert np.array_equal(fxk, outs[0][0]) and np.array_equal(xk, outs[0][1])
        eng.check_unroll_status()
    finally:
        _engine.set_default_engine(old)
