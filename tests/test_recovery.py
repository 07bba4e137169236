"""Recovery from a partner timeout (round 5).

The two-CU unroll (k_unroll_pair) and the persistent MLP unroll (k_mlp_unroll) exchange data between workgroups with
BOUNDED spins: a partner that never shows up (a transient loss of co-residency) raises the workspace's sticky status
word and leaves that launch's outputs invalid.  An evaluation unroll must then be re-run from what it started from on the
exchange-free kernels of the same shapes and return the trajectory it would have returned; a training step must skip its
guarded meta-step (on every rank: tests/test_distributed_cpu.py) and raise.

The timeout is FORCED through the workspace header's fault-injection word (include/l2o_abi.h, ABI v12) -- on the CPU the
oracle-backed engine emulates the same protocol, so the host logic runs in the CPU suite too.  Every trajectory is checked
against the oracle (the reference restated: DM/meta.py:338-389)."""
import warnings

import numpy as np
import pytest

import oracle as O
from helpers import ORACLE_CFGS, make_params, make_problem, max_abs, rel_err
from open_l2o_amd import _abi, _engine, meta, meta_rnnprop_eval, problems
from open_l2o_amd.session import Session
from test_meta_api import _net_config, engine  # noqa: F401  (fixture: oracle engine on CPU, HipEngine under -m gpu)


def _build(name, B, D, T, seed=71):
    cfg = ORACLE_CFGS[name]
    params = make_params(cfg, seed=seed, trained_like=True)
    prob, x0, _ = make_problem("quadratic", B, D, seed=seed + 1)
    problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
    feed = {}
    if cfg.kind == "rnnprop":
        opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
        ml, _, _, step = opt.meta_loss(problem, T)
        feed = {step: 1}
    else:
        opt = meta.MetaOptimizer(**_net_config(cfg, params))
        ml = opt.meta_loss(problem, T)
        step = None
    return cfg, params, prob, x0, opt, ml, feed, step


def _oracle_two_unrolls(cfg, params, prob, x0, B, D, T):
    """fx arrays of two chained T-step unrolls (x, LSTM state and moments carried: MetaLoss.update)."""
    r1 = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), T)
    kw = dict(step0=1 + T)
    if cfg.kind == "rnnprop":
        kw.update(m0=r1.m, v0=r1.v)
    r2 = O.unroll(prob, cfg, params, r1.x, r1.state, T, **kw)
    return r1, r2


@pytest.mark.parametrize("name", ["dm", "rnnprop"])
def test_committed_unroll_recovers_from_injected_timeout(engine, name):
    """Session.run([fx, update]) -- the in-place product path: the second unroll's launch times out (injected), the
    host restores the snapshot of x / LSTM state / moments, re-runs on the exchange-free kernel and returns the
    oracle's trajectory; the third unroll runs on the two-CU kernel again."""
    B, D, T = 8, 32, 12
    cfg, params, prob, x0, opt, ml, feed, step = _build(name, B, D, T)
    graph = opt.graph
    r1, r2 = _oracle_two_unrolls(cfg, params, prob, x0, B, D, T)
    with Session() as sess:
        sess.run(ml.reset)
        fx1 = sess.run([ml.fx, ml.update], feed_dict=feed)[0]
        assert rel_err(np.array([fx1]), r1.fx[-1:]) < 1e-5
        if engine.name != "oracle":
            assert engine.last_unroll_form()[0] == "k_unroll_pair"      # (the shape under test IS the exchanging kernel)
        engine.inject_unroll_fault()
        if step is not None:
            feed[step] = 1 + T
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            fx2, x2, _ = sess.run([ml.fx, ml.x, ml.update], feed_dict=feed)
        assert any("exchange-free" in str(m.message) for m in w), [str(m.message) for m in w]
        assert graph.recoveries == 1
        assert np.isfinite(fx2) and rel_err(np.array([fx2]), r2.fx[-1:]) < 1e-5
        assert max_abs(np.asarray(x2).reshape(r2.x.shape), r2.x) < 1e-5 * max(1.0, float(np.abs(r2.x).max()))
        # the fault was one-shot and nothing is left behind: the next unroll runs the default form and is not recovered
        if step is not None:
            feed[step] = 1 + 2 * T
        fx3 = sess.run([ml.fx, ml.update], feed_dict=feed)[0]
        assert np.isfinite(fx3) and graph.recoveries == 1
        if engine.name != "oracle":
            assert engine.last_unroll_form()[0] == "k_unroll_pair"
    kw = dict(step0=1 + 2 * T)
    if cfg.kind == "rnnprop":
        kw.update(m0=r2.m, v0=r2.v)
    r3 = O.unroll(prob, cfg, params, r2.x, r2.state, T, **kw)
    assert rel_err(np.array([fx3]), r3.fx[-1:]) < 1e-5


def test_eval_epoch_recovers(engine):
    """util.run_eval_epoch's one-long-unroll form (UnrollGraph.execute_many) takes the same recovery."""
    B, D, L, n = 8, 32, 1, 10
    cfg, params, prob, x0, opt, ml, feed, step = _build("dm", B, D, L)
    graph = opt.graph
    ref = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D), n * L)
    graph.reset()
    first = graph.execute_many(n)                            # (also allocates the workspace the fault word lives in)
    assert rel_err(np.array(first), ref.fx[L::L]) < 1e-5 and graph.__dict__.get("recoveries", 0) == 0
    graph.reset()                                            # the same instance again (the problem data is given)
    engine.inject_unroll_fault()
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        vals = graph.execute_many(n)
    assert graph.recoveries == 1
    assert rel_err(np.array(vals), ref.fx[L::L]) < 1e-5


def test_no_recovery_switch_raises(engine, monkeypatch):
    monkeypatch.setenv("L2O_NO_RECOVERY", "1")
    cfg, params, prob, x0, opt, ml, feed, step = _build("dm", 8, 32, 5)
    with Session() as sess:
        sess.run(ml.reset)
        sess.run([ml.fx, ml.update])
        engine.inject_unroll_fault()
        with pytest.raises(_abi.L2OPartnerTimeout):
            sess.run([ml.fx, ml.update])


def test_training_step_skips_its_update_and_raises(engine):
    """A recording unroll that times out leaves a garbage history: the device-side meta-step is guarded by the status
    word and does not run, Adam's step count is taken back, the step raises (a training step is not re-run)."""
    cfg, params, prob, x0, opt0, _, _, _ = _build("dm", 8, 32, 5)
    problem = problems.quadratic(8, 32, data={"w": prob.w, "y": prob.y, "x": x0})
    opt = meta.MetaOptimizer(**_net_config(cfg, params))
    ms = opt.meta_minimize(problem, 5, learning_rate=1e-2)
    with Session() as sess:
        sess.run(ms.reset)
        sess.run([ms.fx, ms.update, ms.step])
        before = opt.save()
        t_before = opt.graph._adam["t"]
        engine.inject_unroll_fault()
        with pytest.raises(_abi.L2OPartnerTimeout):
            sess.run([ms.fx, ms.update, ms.step])
        after = opt.save()
    assert opt.graph._adam["t"] == t_before
    for net in before:
        for mod in before[net]:
            for var in before[net][mod]:
                np.testing.assert_array_equal(np.asarray(after[net][mod][var]), np.asarray(before[net][mod][var]))


@pytest.mark.gpu
def test_mlp_unroll_recovers_from_injected_timeout():
    """config-5 shape (RNNProp on the 784-20-10 MLP, minibatch 64): the persistent MLP unroll times out (injected); the
    host restores the four variables' iterates / LSTM states / moments and re-runs the SAME minibatches on the
    step-granular kernels: equal to the un-faulted persistent launch and to the oracle's multi-variable unroll."""
    from test_mlp_unroll import _sampler
    eng = _engine.HipEngine()
    old = _engine._default_engine
    _engine.set_default_engine(eng)
    try:
        data = problems.synthetic_mnist(512, seed=3)
        T, batch = 30, 64
        idx = np.random.default_rng(72).integers(0, 512, size=(2 * (T + 1), batch))
        cfg = O.RNNPROP
        params = make_params(cfg, seed=73, trained_like=True)
        runs, v0 = [], None
        for fault in (False, True):
            meta.set_random_seed(9)
            problem = problems.mnist(layers=(20,), batch_size=batch, data=data, sampler=_sampler(idx))
            opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
            ml, _, _, step = opt.meta_loss(problem, T)
            graph = opt.graph
            graph.reset()
            v0 = [v.eval() for v in graph.x]
            graph.execute({step: 1}, commit=False)           # (first T + 1 minibatches; allocates the MLP workspace)
            assert graph.last_path == "mlp_unroll"
            if fault:
                eng.inject_unroll_fault()
            with warnings.catch_warnings(record=True):
                warnings.simplefilter("always")
                res = graph.execute({step: 1}, commit=True)  # minibatches T + 1 .. 2 T + 1
            assert graph.__dict__.get("recoveries", 0) == (1 if fault else 0)
            assert graph.last_path == ("steps" if fault else "mlp_unroll")
            runs.append((res["fx_array"], [v.eval() for v in graph.x]))
        ref = O.MnistMLP(data["images"], data["labels"].astype(np.int32), "sigmoid")
        states = [O.net_initial_state(cfg, a.size) for a in v0]
        fx_ref, v_ref, _ = O.unroll_multi(lambda vs, t, wg: ref.fg(vs, idx[T + 1 + t], wg), cfg, params, v0, states, T)
        assert rel_err(runs[0][0], fx_ref) < 1e-5
        assert rel_err(runs[1][0], fx_ref) < 1e-5
        for got, want in zip(runs[1][1], v_ref):
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-6)
    finally:
        _engine.set_default_engine(old)
