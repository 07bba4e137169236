"""MetaOptimizer.meta_minimize(..., second_derivatives=True) (DM/meta.py:328-329: the optimizee gradients are NOT
wrapped in tf.stop_gradient, so the meta-gradient also flows through g_t = grad f(x_t)): the weight gradient of
one training step against torch autograd (float64, create_graph) of the restated unroll, and the reference's own
smoke test (L2O-Swarm/src/meta_test.py:128-138: problems.simple, layers=(), T = 3)."""
import numpy as np
import pytest
import torch

import oracle as O
from helpers import make_params, make_problem
from open_l2o_amd import _engine, meta, meta_rnnprop_eval, problems
from open_l2o_amd.session import Session
from test_meta_api import _net_config


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def hip(request):
    """The product engine on the MI355X (gpu) and -- since round 5 -- the oracle-backed engine on CPU, whose
    vector passes hold their operands to HipEngine's dense-pointer contract: the host logic of the second-derivative
    BPTT (panel grouping, gradient accumulation over groups, the Hessian-vector recursion) runs in the CPU suite."""
    if request.param == "oracle":
        from oracle_engine import OracleEngine
        eng = OracleEngine()
    else:
        eng = _engine.HipEngine()
    old = _engine._default_engine
    _engine.set_default_engine(eng)
    yield eng
    _engine.set_default_engine(old)


def _torch_grad(cfg, params, f, x0, T, second):
    """dL/dtheta of L = sum_t f(x_t), the unroll restated in torch float64; second: keep the graph through g_t."""
    tp = {k: {v: torch.tensor(np.asarray(a, np.float64), requires_grad=True) for v, a in d.items()} for k, d in params.items()}
    x = torch.tensor(x0.astype(np.float64), requires_grad=True)
    n = x.numel()
    H = 20
    st = [[torch.zeros(n, h, dtype=torch.float64), torch.zeros(n, h, dtype=torch.float64)] for h in cfg.layers]

    def cell(inp, h, c, p):
        z = torch.cat([inp, h], 1) @ p["w_gates"] + p["b_gates"]
        i, j, fg, o = torch.sigmoid(z[:, :H]), torch.tanh(z[:, H:2 * H]), torch.sigmoid(z[:, 2 * H:3 * H] + 1), torch.sigmoid(z[:, 3 * H:])
        cn = fg * c + i * j
        return torch.tanh(cn) * o, cn

    loss = 0
    for t in range(T):
        fx = f(x)
        g = torch.autograd.grad(fx, x, create_graph=second, retain_graph=True)[0]
        if not second:
            g = g.detach()
        loss = loss + fx
        gf = g.reshape(-1, 1)
        if cfg.preprocess_name == "LogAndSign":
            eps = float(np.finfo(np.float32).eps)
            a = torch.cat([torch.clamp(torch.log(gf.abs() + eps) / 5, min=-1.0), torch.clamp(gf * float(np.exp(5)), -1.0, 1.0)], 1)
        else:
            a = gf
        out = a
        for li in range(len(cfg.layers)):
            h, c = cell(out, st[li][0], st[li][1], tp["lstm_%d" % (li + 1)])
            st[li] = [h, c]
            out = h
        d = (out @ tp["linear"]["w"] + tp["linear"]["b"]) * cfg.scale
        x = x + d.reshape(x.shape)
    loss = loss + f(x)
    loss.backward()
    return {k: {v: t.grad.numpy() for v, t in d.items()} for k, d in tp.items()}


def _cfg_opts(cfg, params):
    """net config with EVERY option of the oracle NetConfig spelled out (test_meta_api._net_config leaves the scale
    of the identity nets at its default)."""
    opts = {"layers": cfg.layers, "initializer": params, "scale": cfg.scale}
    if cfg.preprocess_name == "LogAndSign":
        opts.update(preprocess_name="LogAndSign", preprocess_options=dict(cfg.preprocess_options))
    return {"cw": {"net": "CoordinateWiseDeepLSTM", "net_options": opts}}


def _captured_grads(opt, ms, feed=None):
    graph = opt.graph
    cap = {}
    orig = graph._adam_apply
    graph._adam_apply = lambda grads, lr, **kw: (cap.update(grads=grads), orig(grads, lr, **kw))[1]
    with Session() as sess:
        sess.run(ms.reset)
        sess.run([ms.fx, ms.update, ms.step], feed_dict=feed or {})
    return {k: np.asarray(v) for k, v in next(iter(cap["grads"].values())).items()}


@pytest.mark.parametrize("kind,pre,scale", [("quadratic", "identity", 0.05), ("rastrigin", "identity", 0.05),
                                            ("quadratic", "LogAndSign", 0.3)])
def test_weight_gradient_with_second_derivatives(hip, kind, pre, scale):
    """Both modes against autograd, and -- the Hessian term moves the gradient by only 0.3-1.5 % here -- the
    DIFFERENCE of the two modes against the difference of the two autograd gradients (5 %)."""
    cfg = O.NetConfig("cw", (20, 20), pre, {"k": 5} if pre == "LogAndSign" else None, scale, False)
    params = make_params(cfg, seed=51)
    B, D, T = 3, 16, 5
    prob, x0, _ = make_problem(kind, B, D, seed=52, stddev=0.3 if kind == "rastrigin" else None)
    if kind == "quadratic":
        W, y = torch.tensor(prob.w.astype(np.float64)), torch.tensor(prob.y.astype(np.float64))

        def f(xx):
            r = torch.matmul(W, xx.unsqueeze(-1)).squeeze(-1) - y
            return torch.mean(torch.sum(r * r, 1))
        xin = x0.reshape(B, -1)
    else:
        A, Bv, C = (torch.tensor(a.astype(np.float64)) for a in (prob.A, prob.B[..., 0], prob.C[..., 0]))

        def f(xx):
            xx = xx.reshape(B, D)
            r = torch.matmul(A, xx.unsqueeze(-1)).squeeze(-1) - Bv
            return torch.mean(0.5 * torch.sum(r * r, 1) - 10.0 * torch.sum(C * torch.cos(2 * np.pi * xx), 1) + 10.0 * D)
        xin = x0
    got, want = {}, {}
    for second in (True, False):
        if kind == "quadratic":
            problem = problems.quadratic(B, D, data={"w": prob.w, "y": prob.y, "x": x0})
        else:
            problem = problems.rastrigin(B, D, data={"A": prob.A, "B": prob.B, "C": prob.C, "x": x0})
        opt = meta.MetaOptimizer(**_cfg_opts(cfg, params))
        ms = opt.meta_minimize(problem, T, learning_rate=1e-6, second_derivatives=second)
        got[second] = _captured_grads(opt, ms)
        want[second] = _torch_grad(cfg, params, f, xin, T, second)
    worst_gap = 0.0
    for (mod, var), g in got[True].items():
        ws, wf = want[True][mod][var].reshape(g.shape), want[False][mod][var].reshape(g.shape)
        scale_g = max(float(np.abs(ws).max()), 1e-12)
        assert float(np.abs(g - ws).max()) / scale_g < 5e-4, (mod, var)
        assert float(np.abs(got[False][(mod, var)] - wf).max()) / scale_g < 5e-4, (mod, var)
        gap = float(np.abs(ws - wf).max())
        worst_gap = max(worst_gap, gap / scale_g)
        if gap / scale_g > 2e-3:                            # blocks where the Hessian term is visible above fp32 noise
            diff_err = float(np.abs((g - got[False][(mod, var)]) - (ws - wf)).max()) / gap
            assert diff_err < 0.05, (mod, var, diff_err)
    assert worst_gap > 2e-3


def test_reference_smoke_simple_problem_linear_net(hip):
    """meta_test.py:128-138 testSecondDerivatives: problems.simple, CoordinateWiseDeepLSTM(layers=()), T = 3 --
    plus the value of the gradient against autograd."""
    cfg = O.NetConfig("cw", (), "identity", None, 1.0, False)
    params = {"linear": {"w": np.array([[-0.3]], np.float32), "b": np.array([0.05], np.float32)}}
    opt = meta.MetaOptimizer(net=dict(net="CoordinateWiseDeepLSTM", net_options={"layers": (), "initializer": params}))
    ms = opt.meta_minimize(problems.simple(), 3, second_derivatives=True)
    got = _captured_grads(opt, ms)
    want = _torch_grad(cfg, params, lambda xx: torch.sum(xx * xx), np.ones((1,), np.float32), 3, True)
    for (mod, var), g in got.items():
        np.testing.assert_allclose(g.reshape(-1), want[mod][var].reshape(-1), rtol=2e-5)
    with Session() as sess:                                   # train(sess, minimize_ops, 1, 2)
        sess.run(ms.reset)
        for _ in range(2):
            cost = sess.run([ms.fx, ms.update, ms.step])[0]
    assert np.isfinite(cost)


def test_second_derivatives_of_a_weighted_term(hip):
    """with a term weight != 1 (problems.ensemble, DM/problems.py:215-245) the recorded gradients carry
    the weight -- so must the Hessian-vector product of the second-order term."""
    cfg = O.NetConfig("cw", (20, 20), "identity", None, 0.05, False)
    params = make_params(cfg, seed=53)
    B, D, T, wgt = 3, 16, 5, 0.5
    prob, x0, _ = make_problem("quadratic", B, D, seed=54)
    W, y = torch.tensor(prob.w.astype(np.float64)), torch.tensor(prob.y.astype(np.float64))

    def f(xx):
        r = torch.matmul(W, xx.unsqueeze(-1)).squeeze(-1) - y
        return wgt * torch.mean(torch.sum(r * r, 1))
    problem = problems.ensemble([{"name": "quadratic", "options": dict(batch_size=B, num_dims=D,
                                                                        data={"w": prob.w, "y": prob.y, "x": x0})}],
                                weights=[wgt])
    opt = meta.MetaOptimizer(**_cfg_opts(cfg, params))
    ms = opt.meta_minimize(problem, T, learning_rate=1e-6, second_derivatives=True)
    got = _captured_grads(opt, ms)
    want = _torch_grad(cfg, params, f, x0.reshape(B, -1), T, True)
    unweighted_hvp = _torch_grad(cfg, params, f, x0.reshape(B, -1), T, False)
    for (mod, var), g in got.items():
        ws = want[mod][var].reshape(g.shape)
        scale_g = max(float(np.abs(ws).max()), 1e-12)
        assert float(np.abs(g - ws).max()) / scale_g < 5e-4, (mod, var)
    assert max(float(np.abs(want[m][v] - unweighted_hvp[m][v]).max()) / max(float(np.abs(want[m][v]).max()), 1e-12)
               for m in want for v in want[m]) > 2e-3        # (the second-order term is visible in this case)


def test_second_derivatives_two_variables_share_a_net(hip):
    """two variables stepped by ONE network with second_derivatives=True go through the BPTT panel by panel,
    so every weight gradient is the sum of two contraction results -- column blocks of two [KA, KB] matrices, i.e.
    strided views that the pointer-taking l2o_lincomb must never see.  A 2-term ensemble (DM/problems.py:215-245) of
    quadratics against autograd of the restated unroll."""
    cfg = O.NetConfig("cw", (20, 20), "identity", None, 0.05, False)
    params = make_params(cfg, seed=55)
    B, D, T = 3, 16, 4
    probs = [make_problem("quadratic", B, D, seed=56 + i) for i in range(2)]
    Ws = [torch.tensor(pr.w.astype(np.float64)) for pr, _, _ in probs]
    ys = [torch.tensor(pr.y.astype(np.float64)) for pr, _, _ in probs]

    def f(xx):                                            # xx: [2, B, D] -- both variables
        tot = 0
        for i in range(2):
            r = torch.matmul(Ws[i], xx[i].unsqueeze(-1)).squeeze(-1) - ys[i]
            tot = tot + torch.mean(torch.sum(r * r, 1))
        return tot
    problem = problems.ensemble([{"name": "quadratic", "options": dict(batch_size=B, num_dims=D,
                                                                        data={"w": pr.w, "y": pr.y, "x": x0})}
                                 for pr, x0, _ in probs])
    x0 = np.stack([x.reshape(B, D) for _, x, _ in probs])
    got, want = {}, {}
    for second in (True, False):
        opt = meta.MetaOptimizer(**_cfg_opts(cfg, params))
        ms = opt.meta_minimize(problem, T, learning_rate=1e-6, second_derivatives=second)
        assert len(opt.graph.x) == 2
        got[second] = _captured_grads(opt, ms)
        want[second] = _torch_grad(cfg, params, f, x0, T, second)
    for (mod, var), g in got[True].items():
        ws, wf = want[True][mod][var].reshape(g.shape), want[False][mod][var].reshape(g.shape)
        scale_g = max(float(np.abs(ws).max()), 1e-12)
        assert float(np.abs(g - ws).max()) / scale_g < 5e-4, (mod, var)
        assert float(np.abs(got[False][(mod, var)] - wf).max()) / scale_g < 5e-4, (mod, var)
