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


def _torch_grad_generic(cfg, params, f, x0, T, step0=1):
    """dL/dtheta of L = sum_t f(x_t) (g_t a constant, DM/meta.py:328-329) for ANY layers tuple, torch float64."""
    import torch
    tp = {k: {v: torch.tensor(np.asarray(a, np.float64), requires_grad=True) for v, a in d.items()} for k, d in params.items()}
    x = torch.tensor(x0.astype(np.float64), requires_grad=True)
    n = x.numel()
    st = [[torch.zeros(n, h, dtype=torch.float64), torch.zeros(n, h, dtype=torch.float64)] for h in cfg.layers]
    m = torch.zeros(n, dtype=torch.float64)
    v = torch.zeros(n, dtype=torch.float64)
    b1 = b2 = float(np.float32(0.95))
    om = float(np.float32(1.0 - 0.95))
    loss = 0
    for t in range(T):
        fx = f(x)
        g = torch.autograd.grad(fx, x, retain_graph=True)[0].detach().reshape(-1)
        loss = loss + fx
        if cfg.kind == "rnnprop":
            k = step0 + t
            m = b1 * m + om * g
            v = b2 * v + om * g * g
            den = torch.sqrt(v / (1.0 - b2 ** k)) + 1e-8
            out = torch.nn.functional.elu(torch.stack([m / (1.0 - b1 ** k) / den, g / den], 1) @ tp["input_projection"]["w"]
                                          + tp["input_projection"]["b"])
        elif cfg.preprocess_name == "LogAndSign":
            eps, kk = float(np.finfo(np.float32).eps), float(cfg.preprocess_options["k"])
            gf = g.reshape(-1, 1)
            out = torch.cat([torch.clamp(torch.log(gf.abs() + eps) / kk, min=-1.0), torch.clamp(gf * float(np.exp(kk)), -1.0, 1.0)], 1)
        else:
            out = g.reshape(-1, 1)
        for li, H in enumerate(cfg.layers):
            p = tp["lstm_%d" % (li + 1)]
            z = torch.cat([out, st[li][0]], 1) @ p["w_gates"] + p["b_gates"]
            i, j, fg, o = torch.sigmoid(z[:, :H]), torch.tanh(z[:, H:2 * H]), torch.sigmoid(z[:, 2 * H:3 * H] + 1), torch.sigmoid(z[:, 3 * H:])
            cn = fg * st[li][1] + i * j
            st[li] = [torch.tanh(cn) * o, cn]
            out = st[li][0]
        d = out @ tp["linear"]["w"] + tp["linear"]["b"]
        if cfg.tanh_output:
            d = torch.tanh(d)
        x = x + (d * cfg.scale).reshape(x.shape)
    loss = loss + f(x)
    loss.backward()
    return {k: {v: t.grad.numpy() for v, t in d.items()} for k, d in tp.items()}


@pytest.mark.parametrize("kind,layers,pre", [("cw", (1,), "identity"), ("cw", (8, 12, 5), "LogAndSign"), ("cw", (64,), "identity"),
                                             ("rnnprop", (16,), "fc"), ("rnnprop", (6, 9), "fc")])
def test_generic_net_meta_gradient_vs_autograd(hip, kind, layers, pre):
    """meta_minimize for ANY `layers` tuple (VERDICT r02 missing #4; DM/networks.py:157 + DM/meta.py:398-414; the
    reference's networks_test.py trains layers=(1,)): the weight gradient of one train step -- l2o_cwlstm_bwd_step_generic
    per step, act^T dz per layer -- against torch autograd of the restated unroll, every block at 5e-4."""
    import torch
    from open_l2o_amd import meta_rnnprop_eval
    popt = {"k": 5} if pre == "LogAndSign" else ({"dim": 12} if pre == "fc" else None)
    cfg = O.NetConfig(kind, layers, pre, popt, 0.05 if kind == "cw" else 0.01, kind == "rnnprop")
    params = make_params(cfg, seed=71)
    B, D, T = 3, 10, 5
    prob, x0, _ = make_problem("quadratic", B, D, seed=72, stddev=0.2)
    W, y = torch.tensor(prob.w.astype(np.float64)), torch.tensor(prob.y.astype(np.float64))

    def f(xx):
        r = torch.matmul(W, xx.unsqueeze(-1)).squeeze(-1) - y
        return torch.mean(torch.sum(r * r, 1))
    problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
    nopts = {"layers": layers, "scale": cfg.scale, "initializer": params}
    feed = {}
    if kind == "rnnprop":
        nopts.update(preprocess_name="fc", preprocess_options=popt, tanh_output=True)
        opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, rp={"net": "RNNprop", "net_options": nopts})
        ms, _, _, step = opt.meta_minimize(problem, T, learning_rate=1e-6)
        feed = {step: 2}
    else:
        if pre == "LogAndSign":
            nopts.update(preprocess_name="LogAndSign", preprocess_options=popt)
        opt = meta.MetaOptimizer(cw={"net": "CoordinateWiseDeepLSTM", "net_options": nopts})
        ms = opt.meta_minimize(problem, T, learning_rate=1e-6)
    graph = opt.graph
    cap = {}
    orig = graph._adam_apply
    graph._adam_apply = lambda grads, lr, **kw: (cap.update(grads=grads), orig(grads, lr, **kw))[1]
    before = {m: {v: np.array(a) for v, a in d.items()} for m, d in next(iter(opt._nets.values())).variables.items()}
    with Session() as sess:
        sess.run(ms.reset)
        sess.run([ms.fx, ms.update, ms.step], feed_dict=feed)
    got = {k: np.asarray(v) for k, v in next(iter(cap["grads"].values())).items()}
    want = _torch_grad_generic(cfg, params, f, x0.reshape(B, -1), T, step0=2 if kind == "rnnprop" else 1)
    assert set(got) == {(m, v) for m in want for v in want[m]}
    for (mod, var), g in got.items():
        w = want[mod][var].reshape(g.shape)
        scale_g = max(float(np.abs(w).max()), 1e-12)
        err = float(np.abs(g - w).max()) / scale_g
        print("layers %r %-16s %-8s |grad|max %.3g rel err %.3g" % (layers, mod, var, scale_g, err))
        assert err < 5e-4, (mod, var, err)
    after = next(iter(opt._nets.values())).variables          # ... and the Adam step moved the weights
    assert any(float(np.abs(after[m][v] - before[m][v]).max()) > 0 for m in before for v in before[m])
