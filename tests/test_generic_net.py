"""The plugin contract of the optimizer networks on the device for configurations outside the harness'
(20, 20) stack: any `layers` tuple through l2o_cwlstm_step_generic (the reference's own tests build layers=(1,)
and (1, 1): L2O-Swarm/src/networks_test.py:33-47), and RNNprop's eager call `net(m, g, prev_state)`
(DM/networks.py:287-295), against the NumPy oracle's net_apply."""
import numpy as np
import pytest

import oracle as O
from helpers import make_params, make_problem, max_abs, rel_err
from open_l2o_amd import _engine, meta, networks, problems
from open_l2o_amd.session import Session

pytestmark = pytest.mark.gpu


@pytest.fixture()
def hip():
    eng = _engine.HipEngine()
    old = _engine._default_engine
    _engine.set_default_engine(eng)
    yield eng
    _engine.set_default_engine(old)


def _np_state(nxt):
    return tuple((h.cpu().numpy(), c.cpu().numpy()) for h, c in nxt.unpack())


@pytest.mark.parametrize("layers,pre", [((1,), "identity"), ((1, 1), "identity"), ((8, 12, 5), "LogAndSign"),
                                       ((64,), "identity"), ((20, 20), "LogAndSign")])
def test_generic_layers_eager_call_vs_oracle(hip, layers, pre):
    """net(g, state) three times in a row (state chained) for several stacks; (20, 20) is included on purpose: it
    runs the matrix-core kernel and must agree with the oracle like the generic kernel does."""
    cfg = O.NetConfig("cw", layers, pre, {"k": 5} if pre == "LogAndSign" else None, 0.1, False)
    params = make_params(cfg, seed=31)
    opts = dict(layers=layers, scale=0.1, initializer=params)
    if pre == "LogAndSign":
        opts.update(preprocess_name="LogAndSign", preprocess_options={"k": 5})
    net = networks.CoordinateWiseDeepLSTM(**opts)
    rng = np.random.default_rng(32)
    shape = (10, 5)                                       # networks_test.py:33 testShape
    state = net.initial_state_for_inputs(hip.tensor(np.zeros(shape)), engine=hip)
    ref_state = O.net_initial_state(cfg, 50)
    for it in range(3):
        g = (rng.standard_normal(shape) * (10.0 ** (it - 1))).astype(np.float32)
        upd, state = net(hip.tensor(g), state)
        want, ref_state = O.net_apply(cfg, params, g, ref_state)
        assert tuple(upd.shape) == shape
        assert max_abs(hip.to_numpy(upd), want) < 2e-6 * max(1.0, float(np.abs(want).max()))
        for (h, c), (hr, cr) in zip(_np_state(state), ref_state):
            assert max_abs(h, hr) < 2e-6 and max_abs(c, cr) < 2e-6 * max(1.0, float(np.abs(cr).max()))


def test_zero_output_layer_gives_exactly_zero_update(hip):
    """networks_test.py:51-69 (testResults) for layers=(1, 1): zero Linear => update == 0 exactly."""
    net = networks.CoordinateWiseDeepLSTM(layers=(1, 1), initializer={"linear": {"w": "zeros", "b": "zeros"}})
    g = hip.tensor(np.random.default_rng(33).standard_normal((10,)))
    upd, _ = net(g, net.initial_state_for_inputs(g, engine=hip))
    assert np.all(hip.to_numpy(upd) == 0)
    assert sum(len(v) for v in networks.CoordinateWiseDeepLSTM(layers=(1,)).variables.values()) == 4   # testTrainable


@pytest.mark.parametrize("layers", [(20, 20), (16,)])
def test_rnnprop_eager_call(hip, layers):
    """RNNprop's plugin contract net(m, g, prev_state) on pre-normalised inputs, chained twice."""
    cfg = O.NetConfig("rnnprop", layers, "fc", {"dim": 20}, 0.01, True)
    params = make_params(cfg, seed=34)
    net = networks.RNNprop(layers=layers, preprocess_name="fc", preprocess_options={"dim": 20}, scale=0.01,
                           tanh_output=True, initializer=params)
    rng = np.random.default_rng(35)
    shape = (7, 9)
    state = net.initial_state_for_inputs(hip.tensor(np.zeros(shape)), engine=hip)
    ref_state = O.net_initial_state(cfg, 63)
    for it in range(2):
        m = rng.standard_normal(shape).astype(np.float32)
        g = rng.standard_normal(shape).astype(np.float32)
        before = None if state.packed is None else state.packed.clone()
        upd, nxt = net(hip.tensor(m), hip.tensor(g), state)
        want, ref_state = O.net_apply(cfg, params, (m, g), ref_state)
        assert tuple(upd.shape) == shape and max_abs(hip.to_numpy(upd), want) < 2e-7 + 2e-5 * float(np.abs(want).max())
        assert bool((state.packed == before).all())      # the arguments are not modified
        state = nxt
        for (h, c), (hr, cr) in zip(_np_state(state), ref_state):
            assert max_abs(h, hr) < 2e-6 and max_abs(c, cr) < 4e-6


def test_rnnprop_eager_call_follows_assign(hip):
    """the eager net(m, g, state) caches a device copy of the weights; assign() (MetaOptimizer.restore,
    the host Adam step) mutates the weight dict IN PLACE -- the next call must run the new weights."""
    layers = (20, 20)
    cfg = O.NetConfig("rnnprop", layers, "fc", {"dim": 20}, 0.01, True)
    params = make_params(cfg, seed=38)
    net = networks.RNNprop(layers=layers, preprocess_name="fc", preprocess_options={"dim": 20}, scale=0.01,
                           tanh_output=True, initializer=params)
    rng = np.random.default_rng(39)
    shape = (5, 16)
    m = rng.standard_normal(shape).astype(np.float32)
    g = rng.standard_normal(shape).astype(np.float32)
    state = net.initial_state_for_inputs(hip.tensor(np.zeros(shape)), engine=hip)
    upd0, _ = net(hip.tensor(m), hip.tensor(g), state)
    new = {k: {v: a.copy() for v, a in d.items()} for k, d in params.items()}
    new["linear"]["w"] = (new["linear"]["w"] * 3.0 + 0.05).astype(np.float32)
    new["lstm_1"]["b_gates"] = (new["lstm_1"]["b_gates"] + 0.2).astype(np.float32)
    for mod in ("linear", "lstm_1"):
        for var, val in new[mod].items():
            net.assign(mod, var, val)
    upd1, _ = net(hip.tensor(m), hip.tensor(g), state)
    want, _ = O.net_apply(cfg, new, (m, g), O.net_initial_state(cfg, 80))
    assert max_abs(hip.to_numpy(upd0), hip.to_numpy(upd1)) > 1e-4          # the weights did change the update
    assert max_abs(hip.to_numpy(upd1), want) < 2e-7 + 2e-5 * float(np.abs(want).max())


def test_generic_net_drives_an_unroll(hip):
    """MetaOptimizer.meta_loss with CoordinateWiseDeepLSTM(layers=(4, 3)) on Quadratic: the step-granular path with the
    generic optimizer step, against the oracle's unroll."""
    cfg = O.NetConfig("cw", (4, 3), "identity", None, 0.1, False)
    params = make_params(cfg, seed=36)
    B, D, T = 3, 10, 6
    prob, x0, _ = make_problem("quadratic", B, D, seed=37)
    meta.set_random_seed(3)
    problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
    opt = meta.MetaOptimizer(cw={"net": "CoordinateWiseDeepLSTM",
                                 "net_options": {"layers": (4, 3), "scale": 0.1, "initializer": params}})
    ml = opt.meta_loss(problem, T)
    with Session() as sess:
        sess.run(ml.reset)
        loss, fx, x, _ = sess.run([ml.loss, ml.fx, ml.x, ml.update])
    assert opt.graph.last_path == "steps"
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T)
    assert rel_err(fx, res.fx[-1]) < 1e-5 and rel_err(loss, res.loss) < 1e-5
    assert max_abs(x[0], res.x.reshape(B, D)) < 1e-5 * max(1.0, float(np.abs(res.x).max()))
    with pytest.raises(NotImplementedError):              # the meta-gradient stays with the (20, 20) / () nets
        ms = opt.meta_minimize(problem, T)
        with Session() as sess:
            sess.run(ms.reset)
            sess.run([ms.fx, ms.update, ms.step])
