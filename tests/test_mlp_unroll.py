"""l2o_mlp_unroll -- the fused persistent unroll of the MLP optimizee (BASELINE config 5: problems.mnist,
784-20-10, minibatch 64, T = 200): parity against the oracle's multi-variable unroll (L2O-DM), against the
step-granular HIP kernels (RNNProp; L2O_OPT_MLP_UNROLL = 0) and the carry / x-scale contracts."""
import numpy as np
import pytest

import oracle as O
from helpers import lib_option, make_params, rel_err
from open_l2o_amd import _abi, _engine, meta, meta_rnnprop_eval, problems
from open_l2o_amd.session import Session
from test_meta_api import _net_config

pytestmark = pytest.mark.gpu


@pytest.fixture()
def hip():
    eng = _engine.HipEngine()
    old = _engine._default_engine
    _engine.set_default_engine(eng)
    yield eng
    _engine.set_default_engine(old)


def _sampler(idx):
    calls = {"n": 0}

    def sampler(n_evals, b, n_data):
        out = idx[calls["n"]:calls["n"] + n_evals]
        calls["n"] += n_evals
        return out
    return sampler


def test_config5_shape_vs_oracle_T200(hip):
    """Minibatch 64, T = 200, the default LogAndSign L2O-DM net on all four variables, ONE launch, against
    O.unroll_multi on the same minibatch sequence: the whole loss trajectory and the final weights."""
    data = problems.synthetic_mnist(512, seed=3)
    T, batch = 200, 64
    idx = np.random.default_rng(70).integers(0, 512, size=(T + 1, batch))
    cfg = O.DM_LOGSIGN
    params = make_params(cfg, seed=71, trained_like=True)
    meta.set_random_seed(9)
    problem = problems.mnist(layers=(20,), batch_size=batch, data=data, sampler=_sampler(idx))
    optimizer = meta.MetaOptimizer(**_net_config(cfg, params))
    ml = optimizer.meta_loss(problem, T)
    with Session() as sess:
        sess.run(ml.reset)
        v0 = [v.eval() for v in optimizer.graph.x]
        fx = sess.run([ml.fx, ml.update])[0]
        fx_array = hip.to_numpy(optimizer.graph._fx_cache[T]["bufs"][0])
        xT = [v.eval() for v in optimizer.graph.x]
    assert optimizer.graph.last_path == "mlp_unroll"
    ref = O.MnistMLP(data["images"], data["labels"].astype(np.int32), "sigmoid")
    states = [O.net_initial_state(cfg, a.size) for a in v0]
    fx_ref, v_ref, _ = O.unroll_multi(lambda vs, t, wg: ref.fg(vs, idx[t], wg), cfg, params, v0, states, T)
    e = rel_err(fx_array, fx_ref)
    print("mlp_unroll T=200 batch=64 vs oracle: rel fx=%.3g fx0=%.5g fx200=%.5g" % (e, fx_ref[0], fx_ref[-1]))
    assert e < 1e-5 and rel_err(fx, fx_ref[-1]) < 1e-5
    for got, want in zip(xT, v_ref):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("activation,batch", [("sigmoid", 64), ("relu", 50), ("sigmoid", 128)])
def test_rnnprop_fused_equals_step_path(hip, activation, batch):
    """RNNProp (Adam moments in registers, bias corrections beta^(step0 + t)) on the MLP optimizee: the fused
    launch against the step-granular kernels on the same minibatches, two chained unrolls (carry of x, LSTM
    state, m, v across launches; step0 = 1 and 1 + T)."""
    data = problems.synthetic_mnist(300, seed=4)
    T = 12
    idx = np.random.default_rng(80).integers(0, 300, size=(2 * (T + 1), batch))
    cfg = O.RNNPROP
    params = make_params(cfg, seed=81, trained_like=True)
    res = {}
    for fused in (1, 0):
        with lib_option(_abi.OPT_MLP_UNROLL, fused):
            meta.set_random_seed(11)
            problem = problems.mnist(layers=(20,), activation=activation, batch_size=batch, data=data, sampler=_sampler(idx))
            opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
            ml, _, _, step = opt.meta_loss(problem, T)
            out = []
            with Session() as sess:
                sess.run(ml.reset)
                for i in range(2):
                    out.append(sess.run([ml.loss, ml.fx, ml.update], feed_dict={step: 1 + i * T})[:2])
                xs = [v.eval() for v in opt.graph.x]
            assert opt.graph.last_path == ("mlp_unroll" if fused else "steps")
            res[fused] = (np.array(out, np.float64), xs)
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=2e-5)
    for a, b in zip(res[1][1], res[0][1]):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("batch", [64, 40])
def test_dm_identity_net_vs_oracle(hip, batch):
    """The plain L2O-DM net (no gradient preprocessing: the kernel's PRE = IDENTITY instantiations, all fragment chunks
    in registers) on the MLP optimizee against O.unroll_multi: minibatch 64 = the FAST instantiation, 40 = the generic one."""
    data = problems.synthetic_mnist(256, seed=6)
    T = 40
    idx = np.random.default_rng(90).integers(0, 256, size=(T + 1, batch))
    cfg = O.DM_IDENTITY
    params = make_params(cfg, seed=91, trained_like=True)
    meta.set_random_seed(13)
    problem = problems.mnist(layers=(20,), batch_size=batch, data=data, sampler=_sampler(idx))
    optimizer = meta.MetaOptimizer(**_net_config(cfg, params))
    ml = optimizer.meta_loss(problem, T)
    with Session() as sess:
        sess.run(ml.reset)
        v0 = [v.eval() for v in optimizer.graph.x]
        sess.run([ml.fx, ml.update])
        fx_array = hip.to_numpy(optimizer.graph._fx_cache[T]["bufs"][0])
        xT = [v.eval() for v in optimizer.graph.x]
    assert optimizer.graph.last_path == "mlp_unroll"
    ref = O.MnistMLP(data["images"], data["labels"].astype(np.int32), "sigmoid")
    states = [O.net_initial_state(cfg, a.size) for a in v0]
    fx_ref, v_ref, _ = O.unroll_multi(lambda vs, t, wg: ref.fg(vs, idx[t], wg), cfg, params, v0, states, T)
    assert rel_err(fx_array, fx_ref) < 1e-5
    for got, want in zip(xT, v_ref):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-6)


def test_unsupported_shapes_fall_back(hip):
    """Hidden width 6 (below the kernel's range): no fused kernel, the step path takes over."""
    data = problems.synthetic_mnist(100, seed=5)
    cfg = O.DM_LOGSIGN
    meta.set_random_seed(12)
    problem = problems.mnist(layers=(6,), batch_size=8, data=data)
    opt = meta.MetaOptimizer(**_net_config(cfg, make_params(cfg, seed=82, trained_like=True)))
    ml = opt.meta_loss(problem, 3)
    with Session() as sess:
        sess.run(ml.reset)
        fx = sess.run([ml.fx, ml.update])[0]
    assert opt.graph.last_path == "steps" and np.isfinite(fx)


def test_hierarchical_allreduce_equals_flat(hip):
    """L2O_OPT_MLP_HIER (round 4, default on): the XCD-hierarchical all-reduce of the hidden pre-activations (per-XCD
    partial sums through L2, ONE fabric hop, 8-way sum in a fixed order) against the flat two-hop protocol on the same
    minibatches -- same losses and weights up to the fp32 summation order of the 245 partial products; and the hierarchical
    run is bit-reproducible (fixed orders everywhere)."""
    data = problems.synthetic_mnist(300, seed=4)
    T, batch = 12, 64
    idx = np.random.default_rng(90).integers(0, 300, size=(2 * (T + 1), batch))
    cfg = O.RNNPROP
    params = make_params(cfg, seed=91, trained_like=True)
    res = {}
    for mode in ("hier", "hier_again", "flat"):
        with lib_option(_abi.OPT_MLP_HIER, 0 if mode == "flat" else 1):
            meta.set_random_seed(13)
            problem = problems.mnist(layers=(20,), batch_size=batch, data=data, sampler=_sampler(idx))
            opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
            ml, _, _, step = opt.meta_loss(problem, T)
            out = []
            with Session() as sess:
                sess.run(ml.reset)
                for i in range(2):
                    out.append(sess.run([ml.loss, ml.fx, ml.update], feed_dict={step: 1 + i * T})[:2])
                xs = [v.eval() for v in opt.graph.x]
            assert opt.graph.last_path == "mlp_unroll"
            res[mode] = (np.array(out, np.float64), xs)
    assert np.array_equal(res["hier"][0], res["hier_again"][0])
    for a, b in zip(res["hier"][1], res["hier_again"][1]):
        assert np.array_equal(a, b)
    np.testing.assert_allclose(res["hier"][0], res["flat"][0], rtol=2e-5)
    for a, b in zip(res["hier"][1], res["flat"][1]):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-6)
