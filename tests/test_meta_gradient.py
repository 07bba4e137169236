"""The meta-gradient path: MetaOptimizer.meta_minimize (DM/meta.py:398-414).

 * the reference's golden test (L2O-Swarm/src/meta_test.py:50-69) run for real: Adam(0.01) on
   the all-zero Linear net over two 5-step unrolls of f(x) = x^2 -> cost 0.7325327, x 0.8559;
 * the hand-derived BPTT step of the oracle (oracle.net_bwd_step) against torch autograd of the
   restated unroll with stop-gradient on the optimizee gradient (float64);
 * the HIP kernel l2o_cwlstm_bwd_step and the whole train step against those (GPU).
"""
import numpy as np
import pytest
import torch

import oracle as O
from helpers import ORACLE_CFGS, lib_option, make_params, make_problem, max_abs, random_state, rel_err, spec_of
from open_l2o_amd import _abi, _engine, meta, meta_rnnprop_eval, problems
from open_l2o_amd.session import Session
from test_meta_api import _net_config, engine  # noqa: F401  (fixture)
from test_oracle_kat import _torch_lstm_from_sonnet  # noqa: F401


def train(sess, minimize_ops, num_epochs, num_unrolls):
    """L2L training, verbatim from meta_test.py:34-44."""
    step, update, reset, loss_last, x_last = minimize_ops
    for _ in range(num_epochs):
        sess.run(reset)
        for _ in range(num_unrolls):
            cost, final_x, unused_1, unused_2 = sess.run([loss_last, x_last, update, step])
    return cost, final_x


def test_results_reference_golden_through_meta_minimize(engine):
    """meta_test.py:50-69 testResults: 'Tests reproducibility of Torch results'."""
    problem = problems.simple()
    optimizer = meta.MetaOptimizer(net=dict(net="CoordinateWiseDeepLSTM",
                                            net_options={"layers": (), "initializer": "zeros"}))
    minimize_ops = optimizer.meta_minimize(problem, 5)
    with Session() as sess:
        cost, final_x = train(sess, minimize_ops, 1, 2)
    torch_cost = 0.7325327
    torch_final_x = 0.8559
    assert abs(float(cost) - torch_cost) < 5e-5                  # assertAlmostEqual(places=4)
    assert abs(float(final_x[0]) - torch_final_x) < 5e-5
    w = optimizer._nets["net"].variables["linear"]
    # two Adam steps of ~-lr each (the cost above was produced with the weights after the FIRST)
    np.testing.assert_allclose([w["w"][0, 0], w["b"][0]], [-0.02, -0.02], rtol=1e-2)


# ------------------------------------------------------------------ autograd reference
def _torch_meta_grad(cfg, params, prob_kind, prob, x0, state0, T, step0=1):
    """dL/dtheta of loss = sum_t f(x_t) by torch autograd (float64) with g detached."""
    tp = {k: {v: torch.tensor(np.asarray(a, np.float64), requires_grad=True) for v, a in d.items()}
          for k, d in params.items()}
    W, y = torch.tensor(prob.w.astype(np.float64)), torch.tensor(prob.y.astype(np.float64))
    B, D = x0.shape
    x = torch.tensor(x0.astype(np.float64))
    st = [[torch.tensor(a.astype(np.float64)) for a in hc] for hc in state0]
    m = torch.zeros_like(x)
    v = torch.zeros_like(x)
    H = 20

    def f(xx):
        r = torch.matmul(W, xx.unsqueeze(-1)).squeeze(-1) - y
        return torch.mean(torch.sum(r * r, 1))

    def cell(inp, h, c, p):
        z = torch.cat([inp, h], 1) @ p["w_gates"] + p["b_gates"]
        i, j, fg, o = torch.sigmoid(z[:, :H]), torch.tanh(z[:, H:2 * H]), torch.sigmoid(z[:, 2 * H:3 * H] + 1), \
            torch.sigmoid(z[:, 3 * H:])
        cn = fg * c + i * j
        return torch.tanh(cn) * o, cn

    loss = 0
    for t in range(T):
        xr = x.detach().clone().requires_grad_(True)
        fx = f(xr)
        g = torch.autograd.grad(fx, xr)[0].detach()
        loss = loss + f(x)
        if cfg.kind == "rnnprop":
            k = float(step0 + t)
            m = 0.95 * m + (1 - 0.95) * g
            v = 0.95 * v + (1 - 0.95) * g * g
            mh, vh = m / (1 - 0.95 ** k), v / (1 - 0.95 ** k)
            feats = torch.stack([(mh / (vh.sqrt() + 1e-8)).reshape(-1), (g / (vh.sqrt() + 1e-8)).reshape(-1)], -1)
            a = torch.nn.functional.elu(feats @ tp["input_projection"]["w"] + tp["input_projection"]["b"])
        elif cfg.preprocess_name == "LogAndSign":
            gf = g.reshape(-1, 1)
            eps = float(np.finfo(np.float64).eps)
            a = torch.cat([torch.clamp(torch.log(gf.abs() + eps) / 5, min=-1.0),
                           torch.clamp(gf * float(np.exp(5)), -1.0, 1.0)], 1)
        else:
            a = g.reshape(-1, 1)
        h1, c1 = cell(a, st[0][0], st[0][1], tp["lstm_1"])
        h2, c2 = cell(h1, st[1][0], st[1][1], tp["lstm_2"])
        st = [[h1, c1], [h2, c2]]
        d = h2 @ tp["linear"]["w"] + tp["linear"]["b"]
        d = (torch.tanh(d) if cfg.tanh_output else d) * cfg.scale
        x = x + d.reshape(x.shape)
    loss = loss + f(x)
    loss.backward()
    return {k: {v: t.grad.numpy() for v, t in d.items()} for k, d in tp.items()}, float(loss.detach())


def _oracle_meta_grad(cfg, params, prob, x0, state0, T, step0=1):
    """Forward with the oracle, backward with oracle.net_bwd_step (float64 when params are)."""
    dt = x0.dtype.type
    x, state = x0.copy(), state0
    m = np.zeros_like(x0)
    v = np.zeros_like(x0)
    hist = []
    for t in range(T):
        g = prob.grad(x)
        if cfg.kind == "rnnprop":
            k = dt(step0 + t)
            m = dt(0.95) * m + dt(1 - 0.95) * g
            v = dt(0.95) * v + dt(1 - 0.95) * g * g
            mh, vh = m / (dt(1) - np.power(dt(0.95), k)), v / (dt(1) - np.power(dt(0.95), k))
            inputs = ((mh / (np.sqrt(vh) + dt(1e-8))).reshape(-1), (g / (np.sqrt(vh) + dt(1e-8))).reshape(-1))
        else:
            inputs = g.reshape(-1)
        hist.append((inputs, state, g))
        delta, state = O.net_apply(cfg, params, inputs if cfg.kind == "rnnprop" else g, state)
        x = x + delta.reshape(x.shape)
    G = prob.grad(x).reshape(-1)
    N = x0.size
    carry = tuple(np.zeros((N, 20), x0.dtype) for _ in range(4))
    grads = {}

    def add(mod, var, val):
        grads.setdefault(mod, {})
        grads[mod][var] = val if var not in grads[mod] else grads[mod][var] + val

    for t in reversed(range(T)):
        inputs, st_prev, g = hist[t]
        carry, rows = O.net_bwd_step(cfg, params, inputs, st_prev, G, carry)
        add("lstm_1", "w_gates", rows["act1"].T @ rows["dz1"])
        add("lstm_1", "b_gates", rows["dz1"].sum(0))
        add("lstm_2", "w_gates", rows["act2"].T @ rows["dz2"])
        add("lstm_2", "b_gates", rows["dz2"].sum(0))
        add("linear", "w", rows["h2"].T @ rows["dd"][:, None])
        add("linear", "b", rows["dd"].sum(keepdims=True))
        if cfg.kind == "rnnprop":
            add("input_projection", "w", rows["feats"].T @ rows["du"])
            add("input_projection", "b", rows["du"].sum(0))
        G = G + g.reshape(-1)
    return grads


@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
def test_oracle_bptt_matches_torch_autograd(name):
    cfg = ORACLE_CFGS[name]
    rng = np.random.default_rng(100)
    params = {k: {v: a.astype(np.float64) for v, a in d.items()} for k, d in make_params(cfg, 101).items()}
    B, D, T = 3, 4, 5
    prob, x0 = O.Quadratic.sample(rng, B, D, stddev=0.5, dtype=np.float64)
    state0 = random_state(cfg, B * D, 102)
    state0 = tuple((h.astype(np.float64), c.astype(np.float64)) for h, c in state0)
    want, _ = _torch_meta_grad(cfg, params, "quadratic", prob, x0, state0, T)
    got = _oracle_meta_grad(cfg, params, prob, x0, state0, T)
    for mod in want:
        for var in want[mod]:
            np.testing.assert_allclose(got[mod][var].reshape(want[mod][var].shape), want[mod][var],
                                       rtol=1e-8, atol=1e-11, err_msg="%s/%s" % (mod, var))


@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
def test_train_step_gradient_and_adam(engine, name):
    """One sess.run([fx, update, step]) == forward + BPTT + Adam: the weights move exactly as
    tf.train.AdamOptimizer would move them given the autograd gradient."""
    cfg = ORACLE_CFGS[name]
    params = make_params(cfg, seed=103, trained_like=True)
    B, D, T = 4, 6, 7
    prob, x0, _ = make_problem("quadratic", B, D, seed=104)
    problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
    key = "rp" if cfg.kind == "rnnprop" else "cw"
    feed = {}
    if cfg.kind == "rnnprop":
        optimizer = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key=key))
        ms, _, _, step_ph = optimizer.meta_minimize(problem, T, learning_rate=0.01)
        feed = {step_ph: 1}
    else:
        optimizer = meta.MetaOptimizer(**_net_config(cfg, params, key=key))
        ms = optimizer.meta_minimize(problem, T, learning_rate=0.01)
    with Session() as sess:
        sess.run(ms.reset)
        cost = sess.run([ms.fx, ms.update, ms.step], feed_dict=feed)[0]
    st0 = O.net_initial_state(cfg, B * D)
    want, loss = _torch_meta_grad(cfg, params, "quadratic", prob, x0, st0, T)
    res = O.unroll(prob, cfg, params, x0, st0, T)
    assert rel_err(cost, res.fx[-1]) < 1e-5 and rel_err(res.loss, loss) < 1e-5
    new = optimizer._nets[key].variables
    for mod in want:
        for var in want[mod]:
            g = want[mod][var].astype(np.float32).reshape(params[mod][var].shape)
            expect, _, _ = O.tf_adam_step(params[mod][var], g, np.zeros_like(g), np.zeros_like(g), 1, lr=0.01)
            # Adam's first step is -lr * sign(g) wherever |g| >> eps: compare where the sign is unambiguous
            mask = np.abs(g) > 1e-5 * np.abs(g).max()
            np.testing.assert_allclose(new[mod][var][mask], expect[mask], rtol=2e-4, atol=1e-7,
                                       err_msg="%s/%s" % (mod, var))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
@pytest.mark.parametrize("B,D", [(3, 10), (2, 37)])
def test_bwd_step_kernel_vs_oracle(name, B, D):
    """l2o_cwlstm_bwd_step (teacher-forced, random state / carries) == oracle.net_bwd_step."""
    eng = _engine.HipEngine()
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=105)
    rng = np.random.default_rng(106)
    N = B * D
    g = rng.standard_normal((B, D)).astype(np.float32)
    state = random_state(cfg, N, 107)
    dx = rng.standard_normal(N).astype(np.float32)
    carry = tuple((rng.standard_normal((N, 20)) * 0.3).astype(np.float32) for _ in range(4))
    m = (rng.standard_normal((B, D)) * 0.1).astype(np.float32)
    v = (rng.random((B, D)) * 0.1 + 0.01).astype(np.float32)
    k = 5
    pw = float(np.float32(0.95)) ** k
    if cfg.kind == "rnnprop":
        dt = np.float32
        mh, vh = m / (dt(1) - dt(pw)), v / (dt(1) - dt(pw))
        inputs = ((mh / (np.sqrt(vh) + dt(1e-8))).reshape(-1), (g / (np.sqrt(vh) + dt(1e-8))).reshape(-1))
    else:
        inputs = g.reshape(-1)
    cout, rows = O.net_bwd_step(cfg, params, inputs, state, dx, carry)
    P = cfg.in_dim
    t = eng.tensor
    io = dict(g=t(g), m=t(m), v=t(v), st_prev=eng.state_pack(*[t(a) for hc in state for a in hc], B, D),
              dx_next=t(dx), carry_in=t(np.stack(carry)), carry_out=eng.zeros(4, N, 20),
              act1=eng.zeros(N, P + 20), dz1=eng.zeros(N, 80), act2=eng.zeros(N, 40), dz2=eng.zeros(N, 80),
              h2=eng.zeros(N, 20), dd=eng.zeros(N), feats=eng.zeros(N, 2), du=eng.zeros(N, 20))
    names = {"w_gates1": ("lstm_1", "w_gates"), "b_gates1": ("lstm_1", "b_gates"), "w_gates2": ("lstm_2", "w_gates"),
             "b_gates2": ("lstm_2", "b_gates"), "w_lin": ("linear", "w"), "b_lin": ("linear", "b"),
             "w_fc": ("input_projection", "w"), "b_fc": ("input_projection", "b")}
    wdev = {kk: t(params[mm][nn]) for kk, (mm, nn) in names.items() if mm in params}
    da = rows.pop("da")
    if cfg.kind != "rnnprop":                  # u = dL/dg of the DM nets (second_derivatives): the preprocess adjoint
        io["dg"] = eng.zeros(N)
        rows["dg"] = O.preprocess_bwd(cfg, g, da)
    eng.bwd_step(spec, wdev, io, pw, pw, B, D)
    for kk, ref in rows.items():
        got = eng.to_numpy(io[kk]).reshape(ref.shape)
        assert max_abs(got, ref) < 2e-5 * max(1.0, float(np.abs(ref).max())), kk
    got = eng.to_numpy(io["carry_out"])
    for i in range(4):
        assert max_abs(got[i], cout[i]) < 2e-5 * max(1.0, float(np.abs(cout[i]).max()))


def test_training_harness_learns(engine, tmp_path):
    """scripts/train_dm.py's schedule (DM/train_dm.py: epochs of truncated-BPTT segments, periodic
    evaluation, best-model .l2l checkpoints) runs and the meta-loss goes down."""
    import argparse
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import _train_common as TC
    flags = argparse.Namespace(save_path=str(tmp_path), num_epochs=12, evaluation_period=4, evaluation_epochs=2,
                               num_steps=20, unroll_length=5, learning_rate=0.01, second_derivatives=False,
                               problem="quadratic", if_scale=True, rd_scale_bound=1.0, if_cl=False, min_num_eval=3,
                               if_mt=False, num_mt=1, seed=11, batch_size=4, num_dims=3)
    tr = TC.Trainer(flags, rnnprop=False)
    w0 = {k: v.copy() for k, v in tr.optimizer._nets["cw"].variables["linear"].items()}
    tr.run()
    w1 = tr.optimizer._nets["cw"].variables["linear"]
    assert not np.array_equal(w0["w"], w1["w"])                      # the meta-step moved the weights
    saved = sorted(os.listdir(str(tmp_path)))
    assert "cw.l2l-0" in saved and any(s.startswith("cw.l2l-") and s != "cw.l2l-0" for s in saved)


@pytest.mark.parametrize("name,B,D", [("dm", 5, 24), ("rnnprop", 5, 24), ("dm", 3, 256), ("rnnprop", 2, 384)])
def test_recording_fused_unroll_equals_step_path(engine, name, B, D, monkeypatch):
    """meta_minimize on a fused-size problem takes the recording unroll (l2o_unroll_record: one
    launch that stores the per-step history) -- the LDS-resident forms for D <= 128, the streaming form
    beyond (config-3 sizes: <= 3 launches per training segment); its meta-gradient == the step-granular path's."""
    cfg = ORACLE_CFGS[name]
    rn = cfg.kind == "rnnprop"
    params = make_params(cfg, seed=71, trained_like=True)
    T = 6
    if D > 128 and engine.name != "hip":
        pytest.skip("the streaming sizes are a property of the HIP kernels")
    prob, x0, _ = make_problem("quadratic", B, D, seed=72)
    got = {}
    for mode in ("fused", "steps"):
        if mode == "steps":
            monkeypatch.setenv("L2O_DISABLE_FUSED", "1")
            monkeypatch.setenv("L2O_BWD_ALIGNED_ONLY", "1")             # (D = 24: the per-step BPTT kernels as reference)
        else:
            monkeypatch.delenv("L2O_DISABLE_FUSED", raising=False)
            monkeypatch.delenv("L2O_BWD_ALIGNED_ONLY", raising=False)
        problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
        if rn:
            opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
            out = opt.meta_minimize(problem, T, learning_rate=1e-3)
            ms, step_ph = out[0], out[3]
        else:
            opt = meta.MetaOptimizer(**_net_config(cfg, params))
            ms, step_ph = opt.meta_minimize(problem, T, learning_rate=1e-3), None
        graph = opt.graph
        cap = {}
        orig = graph._adam_apply
        graph._adam_apply = lambda grads, lr, **kw: (cap.update(grads=grads), orig(grads, lr, **kw))[1]
        with Session() as sess:
            sess.run(ms.reset)
            feed = {step_ph: 3} if rn else {}
            cost = sess.run([ms.fx, ms.update, ms.step], feed_dict=feed)[0]
            x_after = graph.x[0].eval()
        assert graph.last_path == mode
        got[mode] = (cost, x_after, cap["grads"])
    assert rel_err(got["fused"][0], got["steps"][0]) < 1e-5
    assert max_abs(got["fused"][1], got["steps"][1]) < 1e-5
    key = "rp" if rn else "cw"
    for k, gref in got["steps"][2][key].items():
        g = got["fused"][2][key][k]
        scale = max(float(np.abs(gref).max()), 1e-12)
        assert float(np.abs(np.asarray(g) - np.asarray(gref)).max()) / scale < 2e-4, k


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
def test_bwd_tile_kernel_equals_generic_kernel(name, monkeypatch):
    """L2O_OPT_BWD_KERNEL: the matrix-core BPTT (0, the default) and k_cwlstm_bwd_tile (1: tile-aligned panels, LDS-tiled
    I/O, four lanes per coordinate) against the generic one-thread-per-coordinate kernel (2): same meta-gradient."""
    eng = _engine.HipEngine()
    old = _engine._default_engine
    _engine.set_default_engine(eng)
    try:
        cfg = ORACLE_CFGS[name]
        rn = cfg.kind == "rnnprop"
        params = make_params(cfg, seed=81, trained_like=True)
        B, D, T = 3, 32, 4                                   # D % 16 == 0 -> the tile kernel
        prob, x0, _ = make_problem("quadratic", B, D, seed=82)
        got = {}
        for mode in ("mfma", "tile", "generic"):
            _abi.set_option(_abi.OPT_BWD_KERNEL, {"mfma": 0, "tile": 1, "generic": 2}[mode])
            problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
            if rn:
                opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
                out = opt.meta_minimize(problem, T, learning_rate=1e-3)
                ms, step_ph = out[0], out[3]
            else:
                opt = meta.MetaOptimizer(**_net_config(cfg, params))
                ms, step_ph = opt.meta_minimize(problem, T, learning_rate=1e-3), None
            graph = opt.graph
            cap = {}
            orig = graph._adam_apply
            graph._adam_apply = lambda grads, lr, **kw: (cap.update(grads=grads), orig(grads, lr, **kw))[1]
            with Session() as sess:
                sess.run(ms.reset)
                sess.run([ms.fx, ms.update, ms.step], feed_dict={step_ph: 2} if rn else {})
            got[mode] = cap["grads"]["rp" if rn else "cw"]
        for k, gref in got["generic"].items():
            scale = max(float(np.abs(gref).max()), 1e-12)
            for mode in ("mfma", "tile"):
                assert float(np.abs(np.asarray(got[mode][k]) - np.asarray(gref)).max()) / scale < 1e-4, (mode, k)
    finally:
        _abi.set_option(_abi.OPT_BWD_KERNEL, 0)
        _engine.set_default_engine(old)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["tile", "mfma"])
@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
def test_bwd_multi_equals_per_panel_bwd_step(name, form):
    """l2o_cwlstm_bwd_multi (ONE launch for the panels that share a network: the five variables of the
    mnist optimizee, DM/meta.py:231-235) == l2o_cwlstm_bwd_step called once per panel, row for row
    of A / Bm and carry for carry.  Panels: tile-aligned (2 x 32), one flat ragged row (1 x 200),
    one short row (1 x 10).  form == "mfma": the multi-panel launch gets the packed weights and runs
    k_cwlstm_bwd_mfma (bf16x3 matrix-core products, csrc/l2o_bwd_mfma.h); the per-panel reference stays
    on the fp32 tile kernel -- agreement to fp32 rounding (2e-5 of the largest entry) instead of bits."""
    eng = _engine.HipEngine()
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=91, trained_like=True)
    names = {"w_gates1": ("lstm_1", "w_gates"), "b_gates1": ("lstm_1", "b_gates"), "w_gates2": ("lstm_2", "w_gates"),
             "b_gates2": ("lstm_2", "b_gates"), "w_lin": ("linear", "w"), "b_lin": ("linear", "b"),
             "w_fc": ("input_projection", "w"), "b_fc": ("input_projection", "b")}
    t = eng.tensor
    wdev = {kk: t(params[mm][nn]) for kk, (mm, nn) in names.items() if mm in params}
    fc = cfg.kind == "rnnprop"
    P = cfg.in_dim
    K1 = P + 20
    KA = K1 + 60 + (2 if fc else 0) + 1
    KB = 161 + (20 if fc else 0)
    shapes = [(2, 32), (1, 200), (1, 10)]
    rows = [(b * d + 15) // 16 * 16 for b, d in shapes]
    R = sum(rows)
    rng = np.random.default_rng(92)
    cin_all = (rng.standard_normal((4, R, 20)) * 0.3).astype(np.float32)
    segs, ref_A, ref_B, ref_c = [], [], [], []
    row = 0
    for i, (B, D) in enumerate(shapes):
        N = B * D
        g = t(rng.standard_normal((B, D)).astype(np.float32))
        m = t((rng.standard_normal((B, D)) * 0.1).astype(np.float32))
        v = t((rng.random((B, D)) * 0.1 + 0.01).astype(np.float32))
        state = random_state(cfg, N, 93 + i)
        st = eng.state_pack(*[t(a) for hc in state for a in hc], B, D)
        dx = t(rng.standard_normal(N).astype(np.float32))
        segs.append(dict(g=g, m=m if fc else None, v=v if fc else None, st_prev=st, dx_next=dx, B=B, D=D))
        # reference: the single-panel entry point on its own buffers
        A1, B1 = eng.zeros(rows[i], KA), eng.zeros(rows[i], KB)
        co = eng.zeros(4, N, 20)
        io = dict(g=g, m=m if fc else None, v=v if fc else None, st_prev=st, dx_next=dx,
                  carry_in=t(np.ascontiguousarray(cin_all[:, row:row + N])), carry_out=co, a_stride=KA, b_stride=KB,
                  act1=A1[:N, 0:K1], act2=A1[:N, K1:K1 + 40], h2=A1[:N, K1 + 40:K1 + 60],
                  dz1=B1[:N, 0:80], dz2=B1[:N, 80:160], dd=B1[:N, 160:161])
        if fc:
            io.update(feats=A1[:N, K1 + 60:K1 + 62], du=B1[:N, 161:181])
        eng.bwd_step(spec, wdev, io, 0.9, 0.8, B, D)
        ref_A.append(eng.to_numpy(A1)[:N]); ref_B.append(eng.to_numpy(B1)[:N]); ref_c.append(eng.to_numpy(co))
        row += rows[i]
    A, Bm = eng.zeros(R, KA), eng.zeros(R, KB)
    cout = eng.zeros(4, R, 20)
    wmulti = dict(wdev, wpack=eng.pack_weights(spec, params)) if form == "mfma" else wdev
    eng.bwd_multi(spec, wmulti, segs, t(cin_all), cout, A, Bm, 0.9, 0.8)
    A, Bm, cout = eng.to_numpy(A), eng.to_numpy(Bm), eng.to_numpy(cout)

    def same(got, ref, what):
        if form == "tile":
            np.testing.assert_array_equal(got, ref, err_msg=what)
        else:
            # column-wise scale: the columns of A / Bm differ by orders of magnitude
            tol = 2e-5 * np.maximum(np.abs(ref).max(axis=0, keepdims=True), 1e-30) + 1e-9
            assert (np.abs(got - ref) <= tol).all(), (what, float(np.abs(got - ref).max()))

    row = 0
    for i, (B, D) in enumerate(shapes):
        N = B * D
        same(A[row:row + N], ref_A[i], "A panel %d" % i)
        same(Bm[row:row + N], ref_B[i], "Bm panel %d" % i)
        for a in range(4):
            same(cout[a, row:row + N], ref_c[i][a], "carry %d panel %d" % (a, i))
        assert not A[row + N:row + rows[i]].any() and not Bm[row + N:row + rows[i]].any()   # padding rows stay zero
        row += rows[i]


@pytest.mark.gpu
@pytest.mark.parametrize("dx_mode", ["g_final", "table"])
@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
def test_bwd_unroll_equals_stepwise_bwd_multi(name, dx_mode):
    """l2o_cwlstm_bwd_unroll (T steps in one launch, carries in registers, dL/d(delta_t) either from the
    pointer table or accumulated from g_final: loss = sum_t fx_t, DM/meta.py:376) == T calls of
    l2o_cwlstm_bwd_multi on the fp32 tile kernel with the carries going through memory."""
    eng = _engine.HipEngine()
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=95, trained_like=True)
    names = {"w_gates1": ("lstm_1", "w_gates"), "b_gates1": ("lstm_1", "b_gates"), "w_gates2": ("lstm_2", "w_gates"),
             "b_gates2": ("lstm_2", "b_gates"), "w_lin": ("linear", "w"), "b_lin": ("linear", "b"),
             "w_fc": ("input_projection", "w"), "b_fc": ("input_projection", "b")}
    t_ = eng.tensor
    wdev = {kk: t_(params[mm][nn]) for kk, (mm, nn) in names.items() if mm in params}
    wfused = dict(wdev, wpack=eng.pack_weights(spec, params))
    fc = cfg.kind == "rnnprop"
    P = cfg.in_dim
    KA = P + 20 + 60 + (2 if fc else 0) + 1
    KB = 161 + (20 if fc else 0)
    T, step0 = 4, 3
    shapes = [(2, 32), (1, 40)]
    rows = [(b * d + 15) // 16 * 16 for b, d in shapes]
    R = sum(rows)
    rng = np.random.default_rng(96)
    panels = []
    for i, (B, D) in enumerate(shapes):
        N = B * D
        gs = [t_((rng.standard_normal((B, D)) * 0.5).astype(np.float32)) for _ in range(T)]
        ms = [t_((rng.standard_normal((B, D)) * 0.1).astype(np.float32)) for _ in range(T)] if fc else [None] * T
        vs = [t_((rng.random((B, D)) * 0.1 + 0.01).astype(np.float32)) for _ in range(T)] if fc else [None] * T
        sts = []
        for k in range(T):
            state = random_state(cfg, N, 200 + 10 * i + k)
            sts.append(eng.state_pack(*[t_(a) for hc in state for a in hc], B, D))
        g_final = t_((rng.standard_normal(N) * 0.5).astype(np.float32))
        dxs, acc = [None] * T, g_final.clone()
        for k in reversed(range(T)):
            dxs[k] = acc
            acc = acc + gs[k].reshape(N)
        panels.append(dict(B=B, D=D, gs=gs, ms=ms, vs=vs, sts=sts, dxs=dxs, g_final=g_final))
    cin0 = t_((rng.standard_normal((4, R, 20)) * 0.3).astype(np.float32))
    # reference: step by step on the fp32 tile kernel
    A0, B0 = eng.zeros(T, R, KA), eng.zeros(T, R, KB)
    cin, cout = cin0.clone(), eng.zeros(4, R, 20)
    b1, b2 = float(np.float32(spec.beta1)), float(np.float32(spec.beta2))
    for k in reversed(range(T)):
        segs = [dict(g=pn["gs"][k], m=pn["ms"][k], v=pn["vs"][k], st_prev=pn["sts"][k], dx_next=pn["dxs"][k],
                     B=pn["B"], D=pn["D"]) for pn in panels]
        eng.bwd_multi(spec, wdev, segs, cin, cout, A0[k], B0[k], b1 ** (step0 + k), b2 ** (step0 + k))
        cin, cout = cout, cin
    ref_c = eng.to_numpy(cin)
    A1, B1 = eng.zeros(T, R, KA), eng.zeros(T, R, KB)
    c1 = eng.zeros(4, R, 20)
    fused_panels = [dict(pn, dxs=None) if dx_mode == "g_final" else dict(pn, g_final=None) for pn in panels]
    eng.bwd_unroll(spec, wfused, fused_panels, T, step0, A1, B1, carry_in=cin0, carry_out=c1)

    def close(got, ref, what):
        tol = 2e-5 * np.maximum(np.abs(ref).max(axis=0, keepdims=True), 1e-30) + 1e-9
        assert (np.abs(got - ref) <= tol).all(), (what, float(np.abs(got - ref).max()))

    A0, B0, A1, B1, c1 = [eng.to_numpy(x) for x in (A0, B0, A1, B1, c1)]
    for k in range(T):
        close(A1[k], A0[k], "A step %d" % k)
        close(B1[k], B0[k], "Bm step %d" % k)
    row = 0
    for i, (B, D) in enumerate(shapes):                     # (the padding rows of a ragged last tile are not defined)
        for a in range(4):
            close(c1[a, row:row + B * D], ref_c[a, row:row + B * D], "carry %d panel %d" % (a, i))
        row += rows[i]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
def test_bwd_unroll_ragged_problem_tiles(name):
    """l2o_cwlstm_bwd_unroll on panels whose D is NOT a multiple of 16 with B > 1 (BASELINE config 4: d = 100): the
    tiles are per problem, ceil(D / 16) each with a ragged last one -- the packed-state layout the forward records --
    and the rows of A / Bm / the carries are 16 per tile (padding rows zero).  Reference: the one-thread-per-coordinate
    kernel (l2o_cwlstm_bwd_step, any shape), step by step, rows mapped coordinate -> (tile, lane)."""
    eng = _engine.HipEngine()
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=97, trained_like=True)
    names = {"w_gates1": ("lstm_1", "w_gates"), "b_gates1": ("lstm_1", "b_gates"), "w_gates2": ("lstm_2", "w_gates"),
             "b_gates2": ("lstm_2", "b_gates"), "w_lin": ("linear", "w"), "b_lin": ("linear", "b"),
             "w_fc": ("input_projection", "w"), "b_fc": ("input_projection", "b")}
    t_ = eng.tensor
    wdev = {kk: t_(params[mm][nn]) for kk, (mm, nn) in names.items() if mm in params}
    wfused = dict(wdev, wpack=eng.pack_weights(spec, params))
    fc = cfg.kind == "rnnprop"
    P = cfg.in_dim
    K1 = P + 20
    KA = K1 + 60 + (2 if fc else 0) + 1
    KB = 161 + (20 if fc else 0)
    T, step0 = 3, 2
    shapes = [(3, 100), (2, 37), (1, 40), (4, 32)]
    rows = [b * ((d + 15) // 16) * 16 for b, d in shapes]
    R = sum(rows)
    rng = np.random.default_rng(98)
    panels, maps = [], []
    off = 0
    for i, (B, D) in enumerate(shapes):
        N = B * D
        tpp = (D + 15) // 16
        bb, jj = np.divmod(np.arange(N), D)
        maps.append(off + (bb * tpp + jj // 16) * 16 + jj % 16)         # row of coordinate n in the padded numbering
        off += rows[i]
        gs = [t_((rng.standard_normal((B, D)) * 0.5).astype(np.float32)) for _ in range(T)]
        ms = [t_((rng.standard_normal((B, D)) * 0.1).astype(np.float32)) for _ in range(T)] if fc else [None] * T
        vs = [t_((rng.random((B, D)) * 0.1 + 0.01).astype(np.float32)) for _ in range(T)] if fc else [None] * T
        sts = []
        for k in range(T):
            state = random_state(cfg, N, 300 + 10 * i + k)
            sts.append(eng.state_pack(*[t_(a) for hc in state for a in hc], B, D))
        g_final = t_((rng.standard_normal(N) * 0.5).astype(np.float32))
        dxs, acc = [None] * T, g_final.clone()
        for k in reversed(range(T)):
            dxs[k] = acc
            acc = acc + gs[k].reshape(N)
        panels.append(dict(B=B, D=D, gs=gs, ms=ms, vs=vs, sts=sts, dxs=dxs, g_final=g_final))
    cin_rows = (rng.standard_normal((4, R, 20)) * 0.3).astype(np.float32)
    live = np.zeros(R, bool)
    for mp in maps:
        live[mp] = True
    cin_rows[:, ~live] = 0.0
    # reference: per panel, per step, dense [N] rows
    b1, b2 = float(np.float32(spec.beta1)), float(np.float32(spec.beta2))
    refA, refB = np.zeros((T, R, KA), np.float32), np.zeros((T, R, KB), np.float32)
    refC = np.zeros((4, R, 20), np.float32)
    for pn, mp in zip(panels, maps):
        B, D = pn["B"], pn["D"]
        N = B * D
        cin = t_(np.ascontiguousarray(cin_rows[:, mp]))
        for k in reversed(range(T)):
            At, Bt = eng.zeros(N, KA), eng.zeros(N, KB)
            cout = eng.zeros(4, N, 20)
            io = dict(g=pn["gs"][k], dx_next=pn["dxs"][k], st_prev=pn["sts"][k], carry_in=cin, carry_out=cout,
                      m=pn["ms"][k], v=pn["vs"][k], a_stride=KA, b_stride=KB, act1=At[:, 0:K1], act2=At[:, K1:K1 + 40],
                      h2=At[:, K1 + 40:K1 + 60], dz1=Bt[:, 0:80], dz2=Bt[:, 80:160], dd=Bt[:, 160:161])
            if fc:
                io.update(feats=At[:, K1 + 60:K1 + 62], du=Bt[:, 161:181])
            with lib_option(_abi.OPT_BWD_KERNEL, 2):                     # the generic kernel
                eng.bwd_step(spec, wdev, io, b1 ** (step0 + k), b2 ** (step0 + k), B, D)
            a = eng.to_numpy(At)
            a[:, KA - 1] = 1.0
            refA[k, mp], refB[k, mp] = a, eng.to_numpy(Bt)
            cin = cout
        refC[:, mp] = eng.to_numpy(cin)
    for dx_mode in ("g_final", "table"):
        A1, B1 = eng.empty(T, R, KA), eng.empty(T, R, KB)
        A1.fill_(7.0); B1.fill_(7.0)                                     # every row is written, the padding rows as zeros
        c1 = eng.zeros(4, R, 20)
        fused_panels = [dict(pn, dxs=None) if dx_mode == "g_final" else dict(pn, g_final=None) for pn in panels]
        eng.bwd_unroll(spec, wfused, fused_panels, T, step0, A1, B1, carry_in=t_(cin_rows), carry_out=c1)
        A1n, B1n, c1n = eng.to_numpy(A1), eng.to_numpy(B1), eng.to_numpy(c1)
        for k in range(T):
            for got, ref, what in ((A1n[k], refA[k], "A"), (B1n[k], refB[k], "Bm")):
                tol = 2e-5 * np.maximum(np.abs(ref).max(axis=0, keepdims=True), 1e-30) + 1e-9
                assert (np.abs(got - ref) <= tol).all(), (dx_mode, what, k, float(np.abs(got - ref).max()))
            assert not A1n[k][~live].any() and not B1n[k][~live].any()
        for a in range(4):
            tol = 2e-5 * np.maximum(np.abs(refC[a]).max(), 1e-30) + 1e-9
            assert (np.abs(c1n[a][live] - refC[a][live]) <= tol).all(), (dx_mode, "carry", a)


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [16, 64])
@pytest.mark.parametrize("name", ["dm_logsign", "rnnprop"])
def test_planned_mlp_unroll_equals_step_path(name, batch, monkeypatch):
    """The recorded unroll of problems.mnist three ways: ONE persistent launch that also records the history
    (l2o_mlp_unroll_record, the default where the fused MLP unroll applies; batch 64 = its FAST instantiation), the
    step-granular PLAN (history buffers chained through the steps, ctypes arguments prepared once;
    L2O_NO_MLP_UNROLL_RECORD=1) and the plain step-by-step path with cloned history (+ L2O_NO_STEP_PLAN=1): same costs,
    same iterates and the same meta-gradient for two consecutive training steps (the second re-uses the buffers)."""
    eng = _engine.HipEngine()
    old = _engine._default_engine
    _engine.set_default_engine(eng)
    try:
        cfg = ORACLE_CFGS[name]
        rn = cfg.kind == "rnnprop"
        params = make_params(cfg, seed=83, trained_like=True)
        data = problems.synthetic_mnist(200, seed=4)
        T = 3
        idx = np.random.default_rng(5).integers(0, 200, size=(64, batch))

        def sampler(n_evals, b, n_data, _state={"k": 0}):
            k = _state["k"]
            _state["k"] = (k + n_evals) % 32
            return idx[k:k + n_evals, :b]

        got = {}
        for mode in ("fused", "plan", "steps"):
            monkeypatch.delenv("L2O_NO_STEP_PLAN", raising=False)
            monkeypatch.delenv("L2O_NO_MLP_UNROLL_RECORD", raising=False)
            monkeypatch.setenv("L2O_MLP_UNROLL_RECORD_GENERIC", "1")   # (minibatch 16: the kernel's generic loops -- not the default there)
            if mode != "fused":
                monkeypatch.setenv("L2O_NO_MLP_UNROLL_RECORD", "1")
            if mode == "steps":
                monkeypatch.setenv("L2O_NO_STEP_PLAN", "1")
            st = {"k": 0}
            problem = problems.mnist(layers=(20,), batch_size=batch, data=data,
                                     sampler=lambda n, b, nd, _s=st: sampler(n, b, nd, _s))
            meta.set_random_seed(11)
            if rn:
                opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
                out = opt.meta_minimize(problem, T, learning_rate=1e-3)
                ms, step_ph = out[0], out[3]
            else:
                opt = meta.MetaOptimizer(**_net_config(cfg, params))
                ms, step_ph = opt.meta_minimize(problem, T, learning_rate=1e-3), None
            graph = opt.graph
            caps = []
            orig = graph._adam_apply
            graph._adam_apply = lambda grads, lr, **kw: (caps.append(grads), orig(grads, lr, **kw))[1]
            costs = []
            with Session() as sess:
                sess.run(ms.reset)
                for i in range(2):
                    costs.append(sess.run([ms.fx, ms.update, ms.step], feed_dict={step_ph: 1 + i * T} if rn else {})[0])
                xs = [v.eval() for v in graph.x]
            assert ("_step_plan" in graph.__dict__) == (mode == "plan")
            if mode == "fused":
                assert graph.last_path == "mlp_unroll" and "_mlp_record_plan" in graph.__dict__
            else:
                assert graph.last_path == "steps"
            got[mode] = (costs, xs, caps)
        key = "rp" if rn else "cw"
        for mode in ("fused", "plan"):
            for a, b in zip(got[mode][0], got["steps"][0]):
                assert rel_err(a, b) < 2e-6, mode
            for a, b in zip(got[mode][1], got["steps"][1]):
                assert max_abs(a, b) < 2e-6, mode
            for ga, gb in zip(got[mode][2], got["steps"][2]):
                for k, gref in gb[key].items():
                    scale = max(float(np.abs(gref).max()), 1e-12)
                    assert float(np.abs(np.asarray(ga[key][k]) - np.asarray(gref)).max()) / scale < 1e-4, (mode, k)
    finally:
        _engine.set_default_engine(old)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dm", "dm_logsign", "rnnprop"])
def test_device_weight_pack_equals_host_pack(name):
    """l2o_wpack_device (weights that live on the device) == l2o_wpack_host, bit for bit."""
    from open_l2o_amd._engine import HipEngine, pack_weights_host
    from helpers import spec_of
    eng = HipEngine()
    cfg = ORACLE_CFGS[name]
    spec = spec_of(cfg)
    params = make_params(cfg, seed=123, trained_like=True)
    host = pack_weights_host(eng.lib, spec, params)
    names = {"w_gates1": ("lstm_1", "w_gates"), "b_gates1": ("lstm_1", "b_gates"), "w_gates2": ("lstm_2", "w_gates"),
             "b_gates2": ("lstm_2", "b_gates"), "w_lin": ("linear", "w"), "b_lin": ("linear", "b"),
             "w_fc": ("input_projection", "w"), "b_fc": ("input_projection", "b")}
    wdev = {k: eng.tensor(params[m][v]) for k, (m, v) in names.items() if m in params}
    out = eng.tensor(np.full(host.size, 7.0, np.float32))           # stale contents must be overwritten
    eng.pack_weights_device(spec, wdev, out)
    got = eng.to_numpy(out)
    assert np.array_equal(got.view(np.uint32), host.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dm", "rnnprop"])
def test_device_meta_step_equals_host_meta_step(name, monkeypatch):
    """Three meta_minimize steps with Adam + re-pack on the device (l2o_adam_step, l2o_wpack_device) leave the
    same weights, losses and packed copy as the NumPy meta-step (L2O_HOST_ADAM=1); the host dict is refreshed
    lazily, and a host-side assign() after device steps is honoured."""
    from open_l2o_amd import _engine
    eng = _engine.HipEngine()
    old = _engine._default_engine
    _engine.set_default_engine(eng)
    try:
        cfg = ORACLE_CFGS[name]
        rn = cfg.kind == "rnnprop"
        params = make_params(cfg, seed=77, trained_like=True)
        B, D, T = 4, 32, 5
        prob, x0, _ = make_problem("quadratic", B, D, seed=78)
        res = {}
        for mode in ("device", "host"):
            if mode == "host":
                monkeypatch.setenv("L2O_HOST_ADAM", "1")
            else:
                monkeypatch.delenv("L2O_HOST_ADAM", raising=False)
            problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
            if rn:
                opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
                out = opt.meta_minimize(problem, T, learning_rate=1e-2)
                ms, step_ph = out[0], out[3]
            else:
                opt = meta.MetaOptimizer(**_net_config(cfg, params))
                ms, step_ph = opt.meta_minimize(problem, T, learning_rate=1e-2), None
            net = opt.graph.nets["rp" if rn else "cw"]
            costs = []
            with Session() as sess:
                sess.run(ms.reset)
                for i in range(3):
                    feed = {step_ph: 1 + i * T} if rn else {}
                    costs.append(sess.run([ms.fx, ms.update, ms.step], feed_dict=feed)[0])
                assert net._host_stale == (mode == "device")
                w = {m: {v: a.copy() for v, a in d.items()} for m, d in net.variables.items()}
                assert not net._host_stale
                packed = eng.to_numpy(net.wpack(eng)).copy()
                # a host-side write after device steps: picked up by the next unroll
                net.assign("linear", "b", np.array([0.25], np.float32))
                feed = {step_ph: 1 + 3 * T} if rn else {}
                costs.append(sess.run([ms.fx, ms.update, ms.step], feed_dict=feed)[0])
                w2 = {m: {v: a.copy() for v, a in d.items()} for m, d in net.variables.items()}
            res[mode] = (costs, w, packed, w2)
        for a, b in zip(res["device"][0], res["host"][0]):
            assert rel_err(a, b) < 1e-6
        for key in (1, 3):
            for m, d in res["host"][key].items():
                for v, a in d.items():
                    assert max_abs(res["device"][key][m][v], a) <= 1e-7 * max(1.0, float(np.abs(a).max())), (m, v)
        assert not np.array_equal(res["host"][1]["lstm_1"]["w_gates"], params["lstm_1"]["w_gates"])
        # the packed copy the kernels read is the pack of the weights the host sees
        from open_l2o_amd._engine import pack_weights_host
        from helpers import spec_of
        assert np.array_equal(pack_weights_host(eng.lib, spec_of(cfg), res["device"][1]).view(np.uint32),
                              res["device"][2].view(np.uint32))
    finally:
        _engine.set_default_engine(old)


def test_device_meta_step_host_logic_on_cpu(monkeypatch):
    """The host logic of the device meta-step (gradient layout = flat weight buffer, in-place update, lazy
    refresh of the .l2l dict, assign() after device steps) on the oracle-backed engine: same weights as the
    NumPy meta-step, bit for bit (the engine's adam_step is the same NumPy expression)."""
    from oracle_engine import OracleEngine
    cfg = ORACLE_CFGS["rnnprop"]
    params = make_params(cfg, seed=61, trained_like=True)
    B, D, T = 3, 10, 4
    prob, x0, _ = make_problem("quadratic", B, D, seed=62)
    res = {}
    old = _engine._default_engine
    try:
        for mode in ("device", "host"):
            eng = OracleEngine()
            _engine.set_default_engine(eng)
            if mode == "host":
                monkeypatch.setenv("L2O_HOST_ADAM", "1")
            else:
                monkeypatch.delenv("L2O_HOST_ADAM", raising=False)
            opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **_net_config(cfg, params, key="rp"))
            out = opt.meta_minimize(problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0}), T,
                                    learning_rate=1e-2)
            ms, step_ph = out[0], out[3]
            net = opt.graph.nets["rp"]
            with Session() as sess:
                sess.run(ms.reset)
                for i in range(2):
                    sess.run([ms.fx, ms.update, ms.step], feed_dict={step_ph: 1 + i * T})
                assert ("adam_step" in eng.calls) == (mode == "device")
                assert net._host_stale == (mode == "device")
                w = {m: {v: a.copy() for v, a in d.items()} for m, d in net.variables.items()}
                assert not net._host_stale
                net.assign("linear", "b", np.array([0.5], np.float32))
                sess.run([ms.fx, ms.update, ms.step], feed_dict={step_ph: 1 + 2 * T})
                w2 = {m: {v: a.copy() for v, a in d.items()} for m, d in net.variables.items()}
            res[mode] = (w, w2)
    finally:
        _engine.set_default_engine(old)
    for k in (0, 1):
        for m, d in res["host"][k].items():
            for v, a in d.items():
                assert np.array_equal(res["device"][k][m][v], a), (m, v)
    assert not np.array_equal(res["host"][0]["lstm_2"]["w_gates"], params["lstm_2"]["w_gates"])
    assert abs(float(res["host"][1]["linear"]["b"][0]) - 0.5) < 0.05      # the assigned value, moved by one Adam step
