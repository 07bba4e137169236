"""Host-API tests: the reference's MetaOptimizer / networks / problems / util surface.

Every test runs twice: with the oracle-backed engine on CPU (host logic only) and --
marked ``gpu`` -- with the product HipEngine on the MI355X, where it is a parity test
of the whole stack against the oracle.  Mirrors the reference's own tests
(/root/reference/Model_Free_L2O/L2O-Swarm/src/meta_test.py, networks_test.py).
"""
import os

import numpy as np
import pytest

import oracle as O
from helpers import ORACLE_CFGS, make_params, make_problem, rel_err
from open_l2o_amd import (_engine, meta, meta_dm_train, meta_rnnprop_eval, meta_rnnprop_train, networks,
                          problems, util)
from open_l2o_amd.session import Session


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request):
    if request.param == "oracle":
        from oracle_engine import OracleEngine
        eng = OracleEngine()
    else:
        eng = _engine.HipEngine()
    old = _engine._default_engine
    _engine.set_default_engine(eng)
    yield eng
    _engine.set_default_engine(old)


def _net_config(cfg, params, key="cw"):
    opts = {"layers": cfg.layers, "initializer": params}
    if cfg.kind == "rnnprop":
        opts.update(preprocess_name="fc", preprocess_options={"dim": 20}, scale=cfg.scale, tanh_output=True)
        return {key: {"net": "RNNprop", "net_options": opts}}
    if cfg.preprocess_name == "LogAndSign":
        opts.update(preprocess_name="LogAndSign", preprocess_options=dict(cfg.preprocess_options),
                    scale=cfg.scale)
    return {key: {"net": "CoordinateWiseDeepLSTM", "net_options": opts}}


# ------------------------------------------------------------- meta_test.py:50-69
def test_results_golden(engine):
    """Reproducibility of the Torch results quoted by the reference test: cost 0.7325327,
    final_x 0.8559.  Unroll 1 runs the all-zero net ("initializer": "zeros"); Adam's first
    meta-step makes w = b = -0.01 (derivation in tests/test_oracle_kat.py); unroll 2."""
    problem = problems.simple()
    optimizer = meta.MetaOptimizer(net=dict(net="CoordinateWiseDeepLSTM",
                                            net_options={"layers": (), "initializer": "zeros"}))
    loss, update, reset, fx, x = optimizer.meta_loss(problem, 5)
    with Session() as sess:
        sess.run(reset)
        cost, final_x, _ = sess.run([fx, x, update])
        assert cost == 1.0 and final_x[0] == 1.0
        net = optimizer._nets["net"]
        net.assign("linear", "w", np.full((1, 1), -0.01))       # what tf.train.AdamOptimizer(0.01) does
        net.assign("linear", "b", np.full((1,), -0.01))
        cost, final_x, _ = sess.run([fx, x, update])
    assert abs(float(cost) - 0.7325327) < 5e-5                    # assertAlmostEqual(places=4)
    assert abs(float(final_x[0]) - 0.8559) < 5e-5


# ------------------------------------------------------------- meta_test.py:71-126
@pytest.mark.parametrize("net_assignments,net_config", [
    (None, {"net": {"net": "CoordinateWiseDeepLSTM", "net_options": {"layers": (20, 20)}}}),
    ([("net", ["x_0", "x_1"])], {"net": {"net": "CoordinateWiseDeepLSTM", "net_options": {"layers": (20, 20)}}}),
    ([("net1", ["x_0"]), ("net2", ["x_1"])],
     {"net1": {"net": "CoordinateWiseDeepLSTM", "net_options": {"layers": (20, 20)}}, "net2": {"net": "Adam"}}),
    ([("net1", ["x_0"]), ("net2", ["x_0"])],
     {"net1": {"net": "CoordinateWiseDeepLSTM", "net_options": {"layers": (20, 20)}},
      "net2": {"net": "CoordinateWiseDeepLSTM", "net_options": {"layers": ()}}}),
])
def test_multi_optimizer(engine, net_assignments, net_config):
    """Different variable->net mappings in the multi-optimizer problem run and agree with
    a direct oracle evaluation of the same wiring."""
    problem = problems.simple_multi_optimizer(num_dims=2)
    optimizer = meta.MetaOptimizer(**net_config)
    ml = optimizer.meta_loss(problem, 3, net_assignments=net_assignments)
    with Session() as sess:
        sess.run(ml.reset)
        cost, xs, _ = sess.run([ml.fx, ml.x, ml.update])
        cost2, xs2, _ = sess.run([ml.fx, ml.x, ml.update])
    assert np.isfinite(cost) and np.isfinite(cost2)
    assert len(xs) == 2
    # oracle: replay with the same nets
    nets = optimizer._nets
    x = [np.ones((), np.float32), np.ones((), np.float32)]
    assign = net_assignments or [(next(iter(nets)), ["x_0", "x_1"])]
    state = {}
    for t in range(3):
        g = [np.float32(2) * xi for xi in x]
        deltas = [np.float32(0), np.float32(0)]
        for key, names in assign:
            net = nets[key]
            for nm in names:
                j = int(nm[-1])
                sk = (key, j)
                if isinstance(net, networks.Adam):
                    st = state.get(sk, (np.float32(0), np.zeros((1, 1), np.float32), np.zeros((1, 1), np.float32)))
                    d, state[sk] = O.adam_net(g[j].reshape(1), st, 1e-3)
                    deltas[j] = deltas[j] + d[0]
                else:
                    cfg = O.NetConfig("cw", net.spec.layers, "identity", None, 1.0, False)
                    st = state.get(sk, O.net_initial_state(cfg, 1))
                    d, state[sk] = O.net_apply(cfg, net.variables, g[j].reshape(1), st)
                    deltas[j] = deltas[j] + d[0]
        x = [xi + di for xi, di in zip(x, deltas)]
    np.testing.assert_allclose([float(v) for v in xs], [float(v) for v in x], rtol=2e-6)
    assert rel_err(cost, float(x[0]) ** 2 + float(x[1]) ** 2) < 1e-5


def test_default_assignment_needs_single_net(engine):
    optimizer = meta.MetaOptimizer(a={"net": "Adam"}, b={"net": "Sgd"})
    with pytest.raises(ValueError, match="single net config"):
        optimizer.meta_loss(problems.simple(), 2)
    optimizer = meta.MetaOptimizer(a={"net": "Sgd"})
    with pytest.raises(ValueError, match="Repeated netid"):
        optimizer.meta_loss(problems.simple_multi_optimizer(), 2,
                            net_assignments=[("a", ["x_0"]), ("a", ["x_1"])])


# ------------------------------------------------------------- the harness path
@pytest.mark.parametrize("name", ["dm", "dm_logsign"])
@pytest.mark.parametrize("kind", ["quadratic", "lasso", "rastrigin", "square_cos"])
def test_meta_loss_matches_oracle(engine, name, kind):
    """MetaLoss(loss, update, reset, fx, x) on the registry problems == oracle unroll."""
    cfg = ORACLE_CFGS[name]
    params = make_params(cfg, seed=30, trained_like=True)
    B, D, T = 6, 10, 12
    prob, x0, arrays = make_problem(kind, B, D, seed=31)
    if kind == "quadratic":
        problem = problems.quadratic(batch_size=B, num_dims=D, data={"w": prob.w, "y": prob.y, "x": x0})
    elif kind == "lasso":
        problem = problems.lasso(batch_size=B, num_dims=D, l=prob.l, data={"w": prob.w, "y": prob.y, "x": x0})
    elif kind == "square_cos":
        problem = problems.square_cos(batch_size=B, num_dims=D,
                                      data={"w": prob.w, "y": prob.y, "wcos": prob.wcos, "x": x0})
    else:
        problem = problems.rastrigin(batch_size=B, num_dims=D,
                                     data={"A": prob.A, "B": prob.B, "C": prob.C, "x": x0})
    optimizer = meta.MetaOptimizer(**_net_config(cfg, params))
    ml = optimizer.meta_loss(problem, T)
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T)
    res2 = O.unroll(prob, cfg, params, res.x, res.state, T)
    with Session() as sess:
        sess.run(ml.reset)
        # fetching without `update` must not move the variables
        l0, f0 = sess.run([ml.loss, ml.fx])
        l1, f1, x1, _ = sess.run([ml.loss, ml.fx, ml.x, ml.update])
        assert l0 == l1 and f0 == f1
        l2, f2, x2, _ = sess.run([ml.loss, ml.fx, ml.x, ml.update])
    assert optimizer._graph.last_path == "fused"
    assert rel_err(l1, res.loss) < 1e-5 and rel_err(f1, res.fx[-1]) < 1e-5
    assert rel_err(l2, res2.loss) < 2e-5 and rel_err(f2, res2.fx[-1]) < 2e-5
    np.testing.assert_allclose(x1[0], res.x, rtol=1e-4, atol=1e-6)
    assert x1[0].shape == x0.shape


def test_fused_and_step_paths_agree(engine, monkeypatch):
    cfg = O.DM_IDENTITY
    params = make_params(cfg, seed=32, trained_like=True)
    B, D, T = 5, 20, 10
    prob, x0, _ = make_problem("quadratic", B, D, seed=33)
    out = {}
    for mode in ("fused", "steps"):
        if mode == "steps":
            monkeypatch.setenv("L2O_DISABLE_FUSED", "1")
        optimizer = meta.MetaOptimizer(**_net_config(cfg, params))
        ml = optimizer.meta_loss(problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0}), T)
        with Session() as sess:
            sess.run(ml.reset)
            out[mode] = sess.run([ml.loss, ml.fx, ml.update])[:2]
        assert optimizer._graph.last_path == mode
    assert rel_err(out["fused"][0], out["steps"][0]) < 1e-5
    assert rel_err(out["fused"][1], out["steps"][1]) < 1e-5


def test_streaming_size_problem_through_the_api(engine, monkeypatch):
    """A problem beyond the LDS-resident sizes (Lasso 40 x 160 per problem; RNNProp) takes the
    streaming fused unroll through MetaOptimizer.meta_loss and agrees with the oracle and with the
    step-granular path; meta_minimize on it records its history on the step-granular path."""
    cfg = O.RNNPROP
    params = make_params(cfg, seed=34, trained_like=True)
    B, D, M, T = 3, 160, 40, 8
    prob, x0, arrays = make_problem("lasso", B, D, seed=35, M=M)
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, step0=1)
    out = {}
    for mode in ("fused", "steps"):
        if mode == "steps":
            monkeypatch.setenv("L2O_DISABLE_FUSED", "1")
        optimizer = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
        problem = problems.lasso(batch_size=B, num_dims=D, num_rows=M, l=prob.l,
                                 data={"w": prob.w, "y": prob.y, "x": x0})
        ml, _, _, step = optimizer.meta_loss(problem, T)
        with Session() as sess:
            sess.run(ml.reset)
            out[mode] = sess.run([ml.loss, ml.fx, ml.x, ml.update], feed_dict={step: 1})[:3]
        assert optimizer.graph.last_path == mode
        assert rel_err(out[mode][0], res.loss) < 1e-5 and rel_err(out[mode][1], res.fx[-1]) < 1e-5
    np.testing.assert_allclose(out["fused"][2][0], out["steps"][2][0], rtol=1e-4, atol=1e-6)
    monkeypatch.delenv("L2O_DISABLE_FUSED", raising=False)
    optimizer = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
    problem = problems.lasso(batch_size=B, num_dims=D, num_rows=M, l=prob.l, data={"w": prob.w, "y": prob.y, "x": x0})
    mm = optimizer.meta_minimize(problem, T, learning_rate=1e-3)
    ms, step_ph = mm[0], mm[3]
    with Session() as sess:
        sess.run(ms.reset)
        cost = sess.run([ms.fx, ms.update, ms.step], feed_dict={step_ph: 1})[0]
    assert optimizer.graph.last_path == "fused"            # ABI v6: the streaming form records its history too
    assert rel_err(cost, res.fx[-1]) < 1e-5


def test_evaluate_dm_flow(engine):
    """DM/evaluate_dm.py:70-91: util.get_config -> meta_loss(problem, 1) -> reset ->
    run_eval_epoch: the loss record is [f(x_1), ..., f(x_K)]."""
    meta.set_random_seed(5)
    problem, net_config, net_assignments = util.get_config("quadratic",
                                                           problem_options={"batch_size": 8, "num_dims": 6})
    optimizer = meta.MetaOptimizer(**net_config)
    meta_loss = optimizer.meta_loss(problem, 1, net_assignments=net_assignments)
    _, update, reset, cost_op, _ = meta_loss
    K = 7
    with Session() as sess:
        sess.run(reset)
        g = optimizer._graph
        w, y, x0 = g._by_name["w"].eval(), g._by_name["y"].eval(), g._by_name["x"].eval()
        time, cost = util.run_eval_epoch(sess, cost_op, [update], K)
    assert len(cost) == K and time > 0
    cfg = O.DM_IDENTITY
    res = O.unroll(O.Quadratic(w, y), cfg, optimizer._nets["cw"].variables, x0,
                   O.net_initial_state(cfg, x0.size), K)
    assert rel_err(np.array(cost), res.fx[1:]) < 1e-5
    util.print_stats("Epoch 1", sum(cost) / K, time, 1)


def test_reset_resamples_problem(engine):
    meta.set_random_seed(6)
    problem, net_config, _ = util.get_config("quadratic", problem_options={"batch_size": 4, "num_dims": 5})
    optimizer = meta.MetaOptimizer(**net_config)
    ml = optimizer.meta_loss(problem, 2)
    with Session() as sess:
        sess.run(ml.reset)
        w1 = optimizer._graph._by_name["w"].eval()
        t, c = util.run_epoch(sess, ml.fx, [ml.update], ml.reset, 3)
        w2 = optimizer._graph._by_name["w"].eval()
    assert np.isfinite(c) and not np.array_equal(w1, w2)
    assert w1.shape == (4, 5, 5) and w1.min() >= 0 and w1.max() < 1


def test_get_config_registry():
    for name in ("simple", "simple-multi", "quadratic", "rastrigin", "lasso", "square_cos"):
        problem, net_config, _ = util.get_config(name)
        assert callable(problem)
        loss = problem()
        assert len(loss.variables) >= 1
    problem, net_config, _ = util.get_config("quadratic", net_name="RNNprop")
    assert net_config["rp"]["net"] == "RNNprop" and net_config["rp"]["net_options"]["tanh_output"]
    loss = util.get_config("quadratic")[0]()
    assert [v.shape for v in loss.variables] == [(128, 10), (128, 10, 10), (128, 10)]
    assert [v.shape for v in util.get_config("rastrigin")[0]().variables][0] == (128, 2, 1)
    with pytest.raises(ValueError, match="is not a valid problem"):
        util.get_config("no-such-problem")
    with pytest.raises(FileNotFoundError):
        util.get_config("mnist")                       # no dataset offline: must be passed / pointed to
    with pytest.raises(NotImplementedError):
        util.get_config("mnist_conv")
    deeper = util.get_config("mnist_deeper", problem_options={"data": problems.synthetic_mnist(32)})[0]
    assert [v.shape for v in deeper().variables] == [(784, 20), (20,), (20, 20), (20,), (20, 10), (10,)]
    problem, net_config, _ = util.get_config("mnist", problem_options={"data": problems.synthetic_mnist(32)})
    assert [v.name for v in problem().variables] == ["mlp/linear_0/w", "mlp/linear_0/b", "mlp/linear_1/w",
                                                     "mlp/linear_1/b"]
    assert net_config["cw"]["net_options"]["preprocess_name"] == "LogAndSign"
    cfgd = util.get_default_net_config("p")
    assert cfgd["net_options"]["preprocess_options"] == {"k": 5} and cfgd["net_path"] == "p"


# ------------------------------------------------------------- RNNProp
def test_rnnprop_eval_flow(engine):
    """DM/evaluate_rnnprop.py:73-92 with DM/util.py:78-89 feeding step = i + 1."""
    cfg = O.RNNPROP
    params = make_params(cfg, seed=34, trained_like=True)
    B, D, K = 4, 12, 9
    prob, x0, _ = make_problem("lasso", B, D, seed=35)
    problem = problems.lasso(batch_size=B, num_dims=D, l=prob.l, data={"w": prob.w, "y": prob.y, "x": x0})
    optimizer = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
    meta_loss, scale, xvars, step = optimizer.meta_loss(problem, 1)
    _, update, reset, cost_op, _ = meta_loss
    assert len(scale) == 1 and len(xvars) == 1 and step.name == "step"
    with Session() as sess:
        sess.run(reset)
        with pytest.raises(ValueError, match="step"):
            sess.run([cost_op])
        _, cost = util.run_eval_epoch(sess, cost_op, [update], K, step=step, unroll_len=1)
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), K, step0=1)
    assert rel_err(np.array(cost), res.fx[1:]) < 1e-5
    # one 9-step unroll == nine 1-step unrolls
    optimizer2 = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
    ml2, _, _, step2 = optimizer2.meta_loss(problem, K)
    with Session() as sess:
        sess.run(ml2.reset)
        fxK = sess.run([ml2.fx, ml2.update], feed_dict={step2: 1})[0]
    assert rel_err(fxK, cost[-1]) < 1e-5


def test_train_fork_arities_and_scale_placeholder(engine):
    cfg = O.DM_LOGSIGN
    params = make_params(cfg, seed=36, trained_like=True)
    B, D, T = 3, 8, 6
    prob, x0, _ = make_problem("quadratic", B, D, seed=37)
    problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
    optimizer = meta_dm_train.MetaOptimizer(0, **_net_config(cfg, params))
    out = optimizer.meta_loss(problem, T)
    assert len(out) == 10                                          # DM/meta_dm_train.py:526-527
    ml, scale, x, constants, subsets = out[:5]
    assert [c.name for c in constants] == ["w:0", "y:0"] and subsets == [[0]]
    rng = np.random.default_rng(38)
    xs = np.exp(rng.uniform(-1, 1, (B, D))).astype(np.float32)
    with Session() as sess:
        sess.run(ml.reset)
        fx = sess.run([ml.fx, ml.update], feed_dict={scale[0]: xs})[0]
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T, x_scale=xs)
    assert rel_err(fx, res.fx[-1]) < 1e-5
    out = meta_rnnprop_train.MetaOptimizer(0, 0.95, 0.95, **_net_config(O.RNNPROP, make_params(O.RNNPROP, 1),
                                                                        key="rp")).meta_loss(problem, 2)
    assert len(out) == 11
    out = meta_dm_train.MetaOptimizer(1, **_net_config(cfg, params)).meta_loss(problem, T)
    assert len(out) == 10 and [len(o) for o in out[5:]] == [1, 1, 1, 1, 1]   # one imitation task (tests/test_imitation.py)
    out = optimizer.meta_minimize(problem, T, learning_rate=0.01)    # numerics: tests/test_meta_gradient.py
    assert len(out) == 11 and hasattr(out[0], "step")               # DM/meta_dm_train.py:529-558


# ------------------------------------------------------------- meta_test.py:190-236
def test_save_and_restore(engine, tmp_path):
    """Saving and restoring a meta-optimizer (.l2l dill files, networks.save / factory(net_path))."""
    layers = (20, 20)
    networks.set_random_seed(7)
    problem = problems.quadratic(batch_size=4, num_dims=5,
                                 data={k: v for k, v in zip(("w", "y", "x"), _quad_data(4, 5))})
    optimizer = meta.MetaOptimizer(net=dict(net="CoordinateWiseDeepLSTM", net_options={"layers": layers}))
    ml = optimizer.meta_loss(problem, 3)
    with Session() as sess:
        sess.run(ml.reset)
        cost, x, _ = sess.run([ml.fx, ml.x, ml.update])
        result = optimizer.save(sess)
        net_vars = result["net"]
        assert set(net_vars) == {"lstm_1", "lstm_2", "linear"}
        assert net_vars["lstm_1"]["w_gates"].shape == (21, 80) and net_vars["linear"]["w"].shape == (20, 1)
        saved = optimizer.save(sess, path=str(tmp_path))
    net_path = next(iter(saved))
    assert net_path == os.path.join(str(tmp_path), "net.l2l") and os.path.exists(net_path)
    import dill
    on_disk = dill.load(open(net_path, "rb"))
    np.testing.assert_array_equal(on_disk["lstm_2"]["b_gates"], net_vars["lstm_2"]["b_gates"])
    # restore through net_path
    optimizer2 = meta.MetaOptimizer(net=dict(net="CoordinateWiseDeepLSTM", net_options={"layers": layers},
                                             net_path=net_path))
    ml2 = optimizer2.meta_loss(problem, 3)
    with Session() as sess:
        sess.run(ml2.reset)
        cost2, x2, _ = sess.run([ml2.fx, ml2.x, ml2.update])
    assert abs(float(cost) - float(cost2)) <= 1e-3 * abs(float(cost))      # places=3 in the reference
    np.testing.assert_allclose(x[0], x2[0], rtol=1e-6)
    # indexed save + restore() (DM/meta_dm_train.py:257-302)
    opt3 = meta_dm_train.MetaOptimizer(0, net=dict(net="CoordinateWiseDeepLSTM", net_options={"layers": layers}))
    opt3.meta_loss(problem, 3)
    optimizer.save(None, path=str(tmp_path), index=4)
    opt3.restore(None, str(tmp_path), 4)
    np.testing.assert_array_equal(opt3._nets["net"].variables["lstm_1"]["w_gates"], net_vars["lstm_1"]["w_gates"])


def _quad_data(B, D):
    rng = np.random.default_rng(40)
    return (rng.random((B, D, D)).astype(np.float32), rng.random((B, D)).astype(np.float32),
            (rng.standard_normal((B, D)) * 0.01).astype(np.float32))


# ------------------------------------------------------------- networks_test.py
def test_network_shapes_and_zero_init(engine):
    """networks_test.py:29-69: output shape == input shape; zero Linear => update exactly 0."""
    import torch
    for init in ("zeros", {"linear": {"w": "zeros", "b": "zeros"}}, {"linear": "zeros"}):
        net = networks.CoordinateWiseDeepLSTM(layers=(20, 20), initializer=init)
        g = engine.tensor(np.random.default_rng(41).standard_normal((3, 3)))
        state = net.initial_state_for_inputs(g, engine=engine)
        update, nxt = net(g, state)
        assert tuple(update.shape) == (3, 3)
        assert np.all(engine.to_numpy(update) == 0)
        assert len(nxt.unpack()) == 2 and tuple(nxt.unpack()[0][0].shape) == (9, 20)
    net = networks.CoordinateWiseDeepLSTM(layers=(20, 20))
    assert sum(len(v) for v in net.variables.values()) == 6
    lr = 0.25
    sgd = networks.Sgd(learning_rate=lr)
    g = engine.tensor(np.ones((2, 2)))
    upd, _ = sgd(g, sgd.initial_state_for_inputs(g))
    np.testing.assert_allclose(engine.to_numpy(upd), -lr * np.ones((2, 2)))
    adam = networks.Adam(learning_rate=0.0)
    upd, st = adam(g, adam.initial_state_for_inputs(g))
    assert np.all(engine.to_numpy(upd) == 0) and tuple(st[1].shape) == (4, 1)
    if engine.name == "hip":                               # other `layers`: the generic-layers kernel (tests/test_generic_net.py)
        assert networks.CoordinateWiseDeepLSTM(layers=(1,)).wpack(engine).layers == (1,)
    else:
        with pytest.raises(_engine._abi.L2OUnsupported):
            networks.CoordinateWiseDeepLSTM(layers=(1,)).wpack(engine)


# ------------------------------------------------------------- problems.mnist
@pytest.mark.parametrize("activation", ["sigmoid", "relu"])
def test_mnist_mlp_optimizee(engine, activation):
    """BASELINE config 5's optimizee: the 784-20-10 MLP of problems.mnist with the default
    LogAndSign L2O-DM net shared by its four variables, a fresh minibatch per evaluation
    (DM/problems.py:282-286); compared with the oracle's multi-variable unroll."""
    data = problems.synthetic_mnist(300, seed=3)
    T, batch = 6, 32
    rng = np.random.default_rng(70)
    idx = rng.integers(0, 300, size=(2 * (T + 1), batch))
    calls = {"n": 0}

    def sampler(n_evals, b, n_data):
        assert (n_evals, b, n_data) == (T + 1, batch, 300)
        out = idx[calls["n"]:calls["n"] + n_evals]
        calls["n"] += n_evals
        return out

    cfg = O.DM_LOGSIGN
    params = make_params(cfg, seed=71, trained_like=True)
    meta.set_random_seed(9)
    problem = problems.mnist(layers=(20,), activation=activation, batch_size=batch, data=data, sampler=sampler)
    optimizer = meta.MetaOptimizer(**_net_config(cfg, params))
    ml = optimizer.meta_loss(problem, T)
    with Session() as sess:
        sess.run(ml.reset)
        v0 = [v.eval() for v in optimizer.graph.x]
        assert [a.shape for a in v0] == [(784, 20), (20,), (20, 10), (10,)]
        assert 0.005 < v0[0].std() < 0.02                       # _nn_initializers: N(0, 0.01)
        loss1, fx1, x1, _ = sess.run([ml.loss, ml.fx, ml.x, ml.update])
        loss2, fx2, x2, _ = sess.run([ml.loss, ml.fx, ml.x, ml.update])
    # (the HIP engine runs the T steps as ONE persistent launch, l2o_mlp_unroll; the oracle engine per step)
    assert optimizer.graph.last_path == ("mlp_unroll" if engine.name == "hip" else "steps")
    ref = O.MnistMLP(data["images"], data["labels"].astype(np.int32), activation)
    states = [O.net_initial_state(cfg, a.size) for a in v0]
    fx_a, va, sa = O.unroll_multi(lambda vs, t, wg: ref.fg(vs, idx[t], wg), cfg, params, v0, states, T)
    fx_b, vb, _ = O.unroll_multi(lambda vs, t, wg: ref.fg(vs, idx[T + 1 + t], wg), cfg, params, va, sa, T)
    assert rel_err(fx1, fx_a[-1]) < 1e-5 and rel_err(loss1, fx_a.sum()) < 1e-5
    assert rel_err(fx2, fx_b[-1]) < 2e-5 and rel_err(loss2, fx_b.sum()) < 2e-5
    for got, want in zip(x2, vb):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("net", ["dm_logsign", "rnnprop"])
def test_mnist_deeper_optimizee(engine, net):
    """util.get_config("mnist_deeper") (DM/util.py:157-163: problems.mnist(layers=(20, 20)), DM/problems.py:254-288) -- the
    784-20-20-10 MLP, six variables stepped by one shared net, a fresh minibatch per evaluation -- against the oracle's
    multi-variable unroll over two chained unrolls; and the meta-gradient path runs on it (one training step)."""
    data = problems.synthetic_mnist(300, seed=3)
    T, batch = 5, 32
    idx = np.random.default_rng(72).integers(0, 300, size=(2 * (T + 1), batch))
    calls = {"n": 0}

    def sampler(n_evals, b, n_data):
        out = idx[calls["n"]:calls["n"] + n_evals]
        calls["n"] += n_evals
        return out

    cfg = O.DM_LOGSIGN if net == "dm_logsign" else O.RNNPROP
    params = make_params(cfg, seed=73, trained_like=True)
    meta.set_random_seed(9)
    problem, default_cfg, na = util.get_config("mnist_deeper", problem_options={"data": data, "batch_size": batch, "sampler": sampler})
    assert default_cfg["cw"]["net_options"]["preprocess_name"] == "LogAndSign" and na is None
    feeds = [{}, {}]
    if cfg.kind == "rnnprop":
        optimizer = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
        ml, _, _, step = optimizer.meta_loss(problem, T)
        feeds = [{step: 1}, {step: 1 + T}]
    else:
        optimizer = meta.MetaOptimizer(**_net_config(cfg, params))
        ml = optimizer.meta_loss(problem, T)
    with Session() as sess:
        sess.run(ml.reset)
        v0 = [v.eval() for v in optimizer.graph.x]
        assert [a.shape for a in v0] == [(784, 20), (20,), (20, 20), (20,), (20, 10), (10,)]
        loss1, fx1, _ = sess.run([ml.loss, ml.fx, ml.update], feed_dict=feeds[0])
        loss2, fx2, x2, _ = sess.run([ml.loss, ml.fx, ml.x, ml.update], feed_dict=feeds[1])
    assert optimizer.graph.last_path == "steps"
    ref = O.MnistMLP(data["images"], data["labels"].astype(np.int32), "sigmoid")
    states = [O.net_initial_state(cfg, a.size) for a in v0]
    kw = {}
    if cfg.kind == "rnnprop":
        fx_a, va, sa, ma, va2 = O.unroll_multi(lambda vs, t, wg: ref.fg(vs, idx[t], wg), cfg, params, v0, states, T,
                                               return_moments=True)
        fx_b, vb, _ = O.unroll_multi(lambda vs, t, wg: ref.fg(vs, idx[T + 1 + t], wg), cfg, params, va, sa, T, ms=ma, vs=va2,
                                     step0=1 + T)
    else:
        fx_a, va, sa = O.unroll_multi(lambda vs, t, wg: ref.fg(vs, idx[t], wg), cfg, params, v0, states, T)
        fx_b, vb, _ = O.unroll_multi(lambda vs, t, wg: ref.fg(vs, idx[T + 1 + t], wg), cfg, params, va, sa, T)
    assert rel_err(fx1, fx_a[-1]) < 1e-5 and rel_err(loss1, fx_a.sum()) < 1e-5
    assert rel_err(fx2, fx_b[-1]) < 2e-5 and rel_err(loss2, fx_b.sum()) < 2e-5
    for got, want in zip(x2, vb):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-6)
    # the training path (recorded step-granular unroll -> BPTT -> Adam) accepts the six-variable optimizee
    meta.set_random_seed(10)
    opt2 = meta.MetaOptimizer(**_net_config(O.DM_LOGSIGN, make_params(O.DM_LOGSIGN, seed=73, trained_like=True)))
    ms = opt2.meta_minimize(util.get_config("mnist_deeper", problem_options={"data": data, "batch_size": batch})[0], 3,
                            learning_rate=1e-3)
    with Session() as sess:
        sess.run(ms.reset)
        c1 = sess.run([ms.fx, ms.update, ms.step])[0]
        c2 = sess.run([ms.fx, ms.update, ms.step])[0]
    assert np.isfinite(c1) and np.isfinite(c2)


def test_lasso_fixed_shared_matrix(engine):
    """problems.lasso_fixed with ONE [M, N] sensing matrix for the whole batch == the batched form."""
    cfg = O.RNNPROP
    params = make_params(cfg, seed=91, trained_like=True)
    B, Mr, N, T = 4, 6, 10, 5
    rng = np.random.default_rng(92)
    A = rng.random((Mr, N)).astype(np.float32)
    b = rng.random((B, Mr, 1)).astype(np.float32)
    fxs = []
    for dataA in (A, np.broadcast_to(A, (B, Mr, N)).copy()):
        meta_rnnprop_eval.set_random_seed(5)
        problem = problems.lasso_fixed(dataA, b, l=0.1)
        opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
        ml, scale, x, step = opt.meta_loss(problem, T)
        with Session() as sess:
            sess.run(ml.reset)
            fxs.append(sess.run([ml.loss, ml.update], feed_dict={step: 1})[0])
    assert rel_err(fxs[0], fxs[1]) < 1e-6


def test_mnist_mlp_optimizee_with_x_scale(engine):
    """The random rescaling of the train forks (DM/util.py:40-54 feeds exp(U(-b, b)) per coordinate and
    divides the variables by it; DM/meta_dm_train.py:384 evaluates the optimizee at x * scale) on the MLP
    optimizee: f and its gradient w.r.t. x (= scale * grad f(x * scale)) against the oracle's
    multi-variable unroll wrapped the same way, for the plain unroll and for the training step."""
    data = problems.synthetic_mnist(200, seed=5)
    T, batch = 4, 16
    rng = np.random.default_rng(75)
    idx = rng.integers(0, 200, size=(3 * (T + 1), batch))
    calls = {"n": 0}

    def sampler(n_evals, b, n_data):
        out = idx[calls["n"]:calls["n"] + n_evals]
        calls["n"] += n_evals
        return out

    cfg = O.DM_LOGSIGN
    params = make_params(cfg, seed=76, trained_like=True)
    meta.set_random_seed(10)
    problem = problems.mnist(layers=(20,), batch_size=batch, data=data, sampler=sampler)
    optimizer = meta_dm_train.MetaOptimizer(0, **_net_config(cfg, params))
    out = optimizer.meta_minimize(problem, T, learning_rate=1e-3)
    ml, scale = out[0], out[1]
    shapes = [(784, 20), (20,), (20, 10), (10,)]
    assert len(scale) == 4
    scl = [np.exp(rng.uniform(-1, 1, s)).astype(np.float32) for s in shapes]
    feed = {p: v for p, v in zip(scale, scl)}
    with Session() as sess:
        sess.run(ml.reset)
        v0 = [v.eval() for v in optimizer.graph.x]
        fx1, x1, _ = sess.run([ml.fx, ml.x, ml.update], feed_dict=feed)             # plain unroll
        fx2, _, _ = sess.run([ml.fx, ml.update, ml.step], feed_dict=feed)           # training step (recorded unroll)
        x2 = [v.eval() for v in optimizer.graph.x]
    ref = O.MnistMLP(data["images"], data["labels"].astype(np.int32), "sigmoid")

    def fg_scaled(off):
        def fg(vs, t, wg):
            res = ref.fg([v * s for v, s in zip(vs, scl)], idx[off + t], wg)
            if not wg:
                return res
            f, grads = res
            return f, [g * s for g, s in zip(grads, scl)]
        return fg

    states = [O.net_initial_state(cfg, a.size) for a in v0]
    fx_a, va, sa = O.unroll_multi(fg_scaled(0), cfg, params, v0, states, T)
    fx_b, vb, _ = O.unroll_multi(fg_scaled(T + 1), cfg, params, va, sa, T)
    assert rel_err(fx1, fx_a[-1]) < 1e-5
    assert rel_err(fx2, fx_b[-1]) < 2e-5
    for got, want in zip(x1, va):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-6)
    for got, want in zip(x2, vb):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("name", ["dm", "rnnprop"])
def test_eval_epoch_as_one_unroll_equals_the_stepwise_loop(engine, name, monkeypatch):
    """util.run_eval_epoch (DM/util.py:78-89; evaluate_dm.py / evaluate_rnnprop.py call it with
    len_unroll = 1 and num_steps round trips) runs a deterministic optimizee's epoch as ONE unroll:
    same per-unroll losses and same final variables as the sess.run-per-unroll loop
    (L2O_EVAL_STEPWISE=1)."""
    cfg = ORACLE_CFGS[name]
    rn = cfg.kind == "rnnprop"
    params = make_params(cfg, seed=41, trained_like=True)
    B, D, n = 4, 16, 12
    prob, x0, _ = make_problem("quadratic", B, D, seed=42)
    got = {}
    for mode in ("one", "stepwise"):
        if mode == "stepwise":
            monkeypatch.setenv("L2O_EVAL_STEPWISE", "1")
        else:
            monkeypatch.delenv("L2O_EVAL_STEPWISE", raising=False)
        for L in (1, 3):
            problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
            if rn:
                opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
                ml, _, _, step = opt.meta_loss(problem, L)
            else:
                opt = meta.MetaOptimizer(**_net_config(cfg, params))
                ml, step = opt.meta_loss(problem, L), None
            with Session() as sess:
                sess.run(ml.reset)
                calls = len(getattr(opt.graph.engine, "calls", []))
                _, cost = util.run_eval_epoch(sess, ml.fx, [ml.update], n, step=step, unroll_len=L)
                xT = opt.graph.x[0].eval()
                if hasattr(opt.graph.engine, "calls"):     # (the oracle engine logs its entry points)
                    new = opt.graph.engine.calls[calls:]
                    assert (new.count("unroll") == 1) == (mode == "one"), new
            assert len(cost) == n
            got[mode, L] = (np.asarray(cost, np.float64), xT)
    for L in (1, 3):
        np.testing.assert_allclose(got["one", L][0], got["stepwise", L][0], rtol=2e-5)
        np.testing.assert_allclose(got["one", L][1], got["stepwise", L][1], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["dm_logsign", "rnnprop"])
def test_eval_epoch_of_the_sampled_mlp_optimizee(engine, name, monkeypatch):
    """evaluate_*.py --problem mnist: util.run_eval_epoch on the minibatch-sampled MLP optimizee sends the
    epoch's launches without a host round trip per unroll (prepared calls, all index rows drawn up front in
    the loop's order) -- same per-unroll losses and final variables as the sess.run-per-unroll loop."""
    cfg = ORACLE_CFGS[name]
    rn = cfg.kind == "rnnprop"
    params = make_params(cfg, seed=43, trained_like=True)
    data = problems.synthetic_mnist(150, seed=6)
    batch, n = 16, 7
    idx = np.random.default_rng(44).integers(0, 150, size=(64, batch))
    got = {}
    for mode in ("one", "stepwise"):
        if mode == "stepwise":
            monkeypatch.setenv("L2O_EVAL_STEPWISE", "1")
        else:
            monkeypatch.delenv("L2O_EVAL_STEPWISE", raising=False)
        for L in (1, 2):
            st = {"k": 0}

            def sampler(n_evals, b, n_data, _s=st):
                out = idx[_s["k"]:_s["k"] + n_evals, :b]
                _s["k"] += n_evals
                return out

            problem = problems.mnist(layers=(20,), batch_size=batch, data=data, sampler=sampler)
            meta.set_random_seed(12)
            if rn:
                opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
                ml, _, _, step = opt.meta_loss(problem, L)
            else:
                opt = meta.MetaOptimizer(**_net_config(cfg, params))
                ml, step = opt.meta_loss(problem, L), None
            with Session() as sess:
                sess.run(ml.reset)
                _, cost = util.run_eval_epoch(sess, ml.fx, [ml.update], n, step=step, unroll_len=L)
                xT = [v.eval() for v in opt.graph.x]
            assert len(cost) == n and st["k"] == n * (L + 1)
            if isinstance(opt.graph.engine, _engine.HipEngine):
                assert ("_eval_plan" in opt.graph.__dict__) == (mode == "one")
            got[mode, L] = (np.asarray(cost, np.float64), xT)
    for L in (1, 2):
        np.testing.assert_allclose(got["one", L][0], got["stepwise", L][0], rtol=1e-5)
        for a, b in zip(got["one", L][1], got["stepwise", L][1]):
            np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", ["dm", "rnnprop"])
def test_launch_restart_replays_a_prepared_call(engine, name):
    """UnrollGraph.launch(restart=x0) on an unchanged problem instance replays a PREPARED call on the HIP engine (one
    ctypes call instead of rebuilding every argument: the evaluation loops of bench.py would otherwise be host-bound).
    Same numbers as the first (general-path) launch; a swapped problem instance, new weights or `reset` are seen."""
    cfg = ORACLE_CFGS[name]
    params = make_params(cfg, seed=81, trained_like=True)
    B, D, T = 4, 24, 6
    prob, x0, _ = make_problem("quadratic", B, D, seed=82)
    prob2, _, _ = make_problem("quadratic", B, D, seed=83)
    problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
    feed = {}
    if name == "rnnprop":
        opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
        ml, _, _, step = opt.meta_loss(problem, T)
        feed = {step: 1}
    else:
        opt = meta.MetaOptimizer(**_net_config(cfg, params))
        ml = opt.meta_loss(problem, T)
    g = opt.graph
    g.reset()
    x0d = [v.value.clone() for v in g.x]
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T)
    outs = []
    for _ in range(3):                                      # 1st: general path; 2nd, 3rd: the prepared call (HIP)
        fx, xs = g.launch(feed, commit=True, restart=x0d)
        outs.append((engine.to_numpy(fx).copy(), engine.to_numpy(xs[0]).copy()))
    assert rel_err(outs[0][0], res.fx) < 1e-5
    for fxk, xk in outs[1:]:
        assert np.array_equal(fxk, outs[0][0]) and np.array_equal(xk, outs[0][1])
    if hasattr(engine, "prepared_unroll"):
        assert 1 <= len(g.__dict__.get("_fast_unrolls", {})) <= 2        # (the first key predates the packed weights' upload)
    # another problem instance in the same variables (what bench.py's ring does): seen, not replayed from the old one
    g._by_name["w"].value, g._by_name["y"].value = engine.tensor(prob2.w), engine.tensor(prob2.y)
    res2 = O.unroll(prob2, cfg, params, x0, O.net_initial_state(cfg, B * D), T)
    for _ in range(2):
        fx, _ = g.launch(feed, commit=True, restart=x0d)
        assert rel_err(engine.to_numpy(fx), res2.fx) < 1e-5
    # new weights
    net = next(iter(opt._nets.values()))
    p2 = {m: {v: a.copy() for v, a in d.items()} for m, d in params.items()}
    p2["linear"]["w"] = (p2["linear"]["w"] * 0.5).astype(np.float32)
    for var, val in p2["linear"].items():
        net.assign("linear", var, val)
    res3 = O.unroll(prob2, cfg, p2, x0, O.net_initial_state(cfg, B * D), T)
    for _ in range(2):
        fx, _ = g.launch(feed, commit=True, restart=x0d)
        assert rel_err(engine.to_numpy(fx), res3.fx) < 1e-5


@pytest.mark.parametrize("name", ["dm", "rnnprop"])
def test_session_run_replays_a_prepared_call(engine, name):
    """The PRODUCT path -- sess.run([fx, update]) in a loop, what evaluate_*.py does (DM/evaluate_dm.py:88-91) -- goes
    through the same prepared-call cache as bench.py's restart= launches (the headline must be what an API
    user gets): five consecutive committed unrolls of L steps equal ONE oracle unroll of 5 L steps (x, LSTM state and
    RNNProp's moments carry; RNNProp's fed step is a call-time argument of the prepared call); Variable.load and a new
    engine workspace layout drop / invalidate the cache."""
    cfg = ORACLE_CFGS[name]
    params = make_params(cfg, seed=91, trained_like=True)
    B, D, L, n = 4, 24, 3, 5
    prob, x0, _ = make_problem("quadratic", B, D, seed=92)
    problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
    if name == "rnnprop":
        opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
        ml, _, _, step = opt.meta_loss(problem, L)
    else:
        opt = meta.MetaOptimizer(**_net_config(cfg, params))
        ml, step = opt.meta_loss(problem, L), None
    res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), n * L)
    g = opt.graph
    with Session() as sess:
        sess.run(ml.reset)
        got = [sess.run([ml.fx, ml.update], feed_dict={} if step is None else {step: 1 + k * L})[0] for k in range(n)]
        xT = g.x[0].eval()
        assert rel_err(got, [res.fx[(k + 1) * L] for k in range(n)]) < 1e-5
        np.testing.assert_allclose(xT.reshape(B, D), np.asarray(res.x).reshape(B, D), rtol=1e-4, atol=1e-6)
        if hasattr(engine, "prepared_unroll"):
            assert len(g.__dict__.get("_fast_unrolls", {})) == 1            # one entry serves every step0
            # another graph re-initialises the shared workspace for a different layout: the stale closure must say so
            ent = next(iter(g._fast_unrolls.values()))
            engine._ws_layout = -12345
            assert ent["call"](g._fx_cache[L]["bufs"][0], 1) is False
            engine._ws_layout = None                                         # (the next general launch re-initialises)
            g.x[0].load(x0)
            assert "_fast_unrolls" not in g.__dict__


@pytest.mark.parametrize("kind,B,D", [("quadratic", 8, 16), ("quadratic", 128, 128)])
def test_run_epoch_defers_all_but_the_last_loss(engine, monkeypatch, kind, B, D):
    """util.run_epoch returns the cost of the LAST unroll only (DM/util.py:75): the meta-training steps before it are
    enqueued without a host sync (Session.run(_defer_loss=True): guarded device-side Adam behind the unroll).  Same
    weights, same cost, bit for bit, as the epoch with one sync per unroll (L2O_NO_DEFER=1)."""
    out = {}
    for mode in ("sync", "defer"):
        if mode == "sync":
            monkeypatch.setenv("L2O_NO_DEFER", "1")
        else:
            monkeypatch.delenv("L2O_NO_DEFER", raising=False)
        meta.set_random_seed(21)
        np.random.seed(3)
        problem, net_config, _ = util.get_config(kind, problem_options={"batch_size": B, "num_dims": D})
        opt = meta.MetaOptimizer(**net_config)
        ms = opt.meta_minimize(problem, 5, learning_rate=1e-3)
        syncs = []
        orig = engine.to_numpy
        monkeypatch.setattr(engine, "to_numpy", lambda t, _o=orig: (syncs.append(1), _o(t))[1])
        with Session() as sess:
            costs = [util.run_epoch(sess, ms.fx, [ms.update, ms.step], ms.reset, 4)[1] for _ in range(2)]
        monkeypatch.setattr(engine, "to_numpy", orig)
        key = next(iter(opt._nets))
        out[mode] = (costs, {m: {v: np.array(a) for v, a in d.items()} for m, d in opt._nets[key].variables.items()},
                     len(syncs), opt.graph.__dict__["_adam"]["t"])
    assert out["sync"][0] == out["defer"][0] and all(np.isfinite(c) for c in out["defer"][0])
    for m, d in out["sync"][1].items():
        for v, a in d.items():
            assert np.array_equal(a, out["defer"][1][m][v]), (m, v)
    assert out["sync"][3] == out["defer"][3] == 8
    assert out["defer"][2] < out["sync"][2]                   # fewer host round trips


@pytest.mark.gpu
def test_deferred_train_steps_report_a_failed_unroll_and_leave_the_weights_alone():
    """A partner timeout of a fused unroll (injected here: the sticky status word of the workspace set by hand) while
    training steps are being enqueued without a host sync: the guarded device-side Adam of every later step is skipped,
    the next synchronous step raises, the weights are what they were before the failure."""
    import torch
    eng = _engine.HipEngine()
    old = _engine._default_engine
    _engine.set_default_engine(eng)
    try:
        meta.set_random_seed(5)
        problem, net_config, _ = util.get_config("quadratic", problem_options={"batch_size": 128, "num_dims": 128})
        opt = meta.MetaOptimizer(**net_config)
        ms = opt.meta_minimize(problem, 5, learning_rate=1e-3)
        key = next(iter(opt._nets))
        with Session() as sess:
            sess.run(ms.reset)
            sess.run([ms.fx, ms.update, ms.step])
            assert opt.graph.last_path == "fused"
            sess.run([ms.fx, ms.update, ms.step], _defer_loss=True)          # runs: the status is clean
            w_ok = {m: {v: np.array(a) for v, a in d.items()} for m, d in opt._nets[key].variables.items()}
            t_ok = opt.graph.__dict__["_adam"]["t"]
            eng._last_ws[:4] = torch.tensor([1, 0, 0, 0], dtype=torch.uint8)
            sess.run([ms.fx, ms.update, ms.step], _defer_loss=True)          # enqueued, but its Adam is guarded off
            with pytest.raises(Exception) as ei:
                sess.run([ms.fx, ms.update, ms.step])
            assert "partner" in str(ei.value) or "timeout" in str(ei.value).lower() or "status" in str(ei.value).lower()
        w_after = opt._nets[key].variables
        for m, d in w_ok.items():
            for v, a in d.items():
                assert np.array_equal(a, np.asarray(w_after[m][v])), (m, v)
        assert opt.graph.__dict__["_adam"]["t"] <= t_ok                       # the skipped steps are not counted
        with Session() as sess:                                              # the status was cleared by the check: training resumes
            sess.run(ms.reset)
            c = sess.run([ms.fx, ms.update, ms.step])[0]
        assert np.isfinite(c)
    finally:
        _engine.set_default_engine(old)


# ------------------------------------------------------------- bench.py --emulate-world (round 5)
def test_emulated_shards_add_up_to_the_global_batch(engine):
    """_graph_core.emulate_world(rank, world): one shard of a sharded job in a single process -- contiguous batch slice,
    1/B_global in every gradient, NO collective.  The shards' partial losses add up to the one-process run on the global
    batch (what the all-reduce of SURVEY.md 8e would have produced), and every shard's iterates are the corresponding
    rows of the global run."""
    from open_l2o_amd import _graph_core
    cfg = ORACLE_CFGS["dm"]
    params = make_params(cfg, seed=90, trained_like=True)
    B, D, T, world = 8, 32, 6, 4
    prob, x0, _ = make_problem("rastrigin", B, D, seed=91)

    def run():
        problem = problems.rastrigin(B, D, data={"A": prob.A, "B": prob.B, "C": prob.C, "x": x0})
        opt = meta.MetaOptimizer(**_net_config(cfg, params))
        ml = opt.meta_loss(problem, T)
        opt.graph.reset()
        res = opt.graph.execute({}, commit=True)
        return opt.graph, res["fx_array"], np.asarray(res["x"][0])
    graph, fx_full, x_full = run()
    assert not graph.sharded
    total = np.zeros_like(fx_full)
    try:
        for r in range(world):
            _graph_core.emulate_world(r, world)
            g, fx_r, x_r = run()
            assert g.sharded and g.shard == (r * B // world, (r + 1) * B // world) and x_r.shape[0] == B // world
            np.testing.assert_allclose(x_r.reshape(B // world, -1), x_full.reshape(B, -1)[r * B // world:(r + 1) * B // world],
                                       rtol=1e-5, atol=1e-6)
            total += fx_r
    finally:
        _graph_core.emulate_world()
    np.testing.assert_allclose(total, fx_full, rtol=2e-6)
    with pytest.raises(ValueError):
        _graph_core.emulate_world(4, 4)


# ------------------------------------------------------------- replicas.Replicas (round 6)
def test_replicas_share_one_network_and_equal_separate_graphs(engine):
    """Replicas: N optimizee instances stepped by ONE optimizer, run together.  On every engine the result equals N separate
    meta_loss graphs run one after the other (same seeds: same initial weights, same minibatches); the HIP engine takes
    the one-instance-per-XCD kernel for the reference's shape."""
    from open_l2o_amd.replicas import Replicas
    data = problems.synthetic_mnist(200, seed=8)
    cfg = ORACLE_CFGS["rnnprop"]
    params = make_params(cfg, seed=83, trained_like=True)
    T, n = 5, 3
    idxs = [np.random.default_rng(100 + j).integers(0, 200, size=(2 * (T + 1), 64)) for j in range(n)]

    def sampler_of(ix):
        calls = {"n": 0}

        def sampler(n_evals, b, n_data):
            out = ix[calls["n"]:calls["n"] + n_evals]
            calls["n"] += n_evals
            return out
        return sampler

    def probs():
        return [problems.mnist(layers=(20,), batch_size=64, data=data, sampler=sampler_of(ix)) for ix in idxs]

    meta.set_random_seed(21)
    reps = Replicas(meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp")), probs(), T)
    assert all(g.nets is reps.graphs[0].nets for g in reps.graphs)
    reps.reset()
    got = [reps.run({reps.step: 1 + i * T}) for i in range(2)]
    assert reps.last_form == ("xcd" if engine.name == "hip" else "chip")
    # the same three instances as three separate graphs
    meta.set_random_seed(21)
    want = []
    graphs = []
    for p in probs():
        opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
        ml, _, _, step = opt.meta_loss(p, T)
        graphs.append((opt, ml, step))
    with Session() as sess:
        for _, ml, _ in graphs:
            sess.run(ml.reset)
        for i in range(2):
            want.append([sess.run([ml.fx, ml.update], feed_dict={step: 1 + i * T})[0] for _, ml, step in graphs])
    np.testing.assert_allclose(np.array(got), np.array(want), rtol=3e-5)
