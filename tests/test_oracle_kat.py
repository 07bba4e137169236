"""Pins the oracle against every known-answer test the reference holds for the
hot path and against independent restatements
(torch.nn.LSTMCell for the Sonnet cell, torch autograd for the gradients).

Reference tests mirrored here (all under
/root/reference/Model_Free_L2O/L2O-Swarm/src/):
  meta_test.py:50-69       cost 0.7325327, final_x 0.8559
  problems_test.py:44-51   simple: f == x^2
  problems_test.py:72-79   simple_multi_optimizer
  problems_test.py:99-111  quadratic(batch=1,dims=1): (w x - y)^2
  preprocess_test.py:37-65 Clamp min/max
  preprocess_test.py:68-98 LogAndSign shape / log(1)=0 / sign column
  networks_test.py:51-69   zero Linear => update exactly 0
  networks_test.py:140-151 Sgd update == -lr*g
  networks_test.py:154-185 Adam shapes, lr=0 => 0
"""
import numpy as np
import pytest
import torch

import oracle as O


# ---------------------------------------------------------------- meta_test KAT
def _simple_unrolls(w, b, n_unrolls, T=5):
    cfg = O.NetConfig("cw", (), "identity", None, 1.0, False)
    params = {"linear": {"w": np.full((1, 1), w, np.float32),
                         "b": np.full((1,), b, np.float32)}}
    prob = O.Simple()
    x = prob.init_x()
    res = None
    for _ in range(n_unrolls):
        res = O.unroll(prob, cfg, params, x, (), T)
        x = res.x
    return res


def test_meta_test_golden_cost_and_x():
    # unroll 1 (zero net): delta == 0, x stays 1 -> Adam's first step is
    # -lr * g/(|g|+eps) = -0.01 on both w and b (gradient signs derived below)
    r1 = _simple_unrolls(0.0, 0.0, 1)
    assert r1.x == np.float32(1.0)
    assert np.all(r1.fx == np.float32(1.0))
    # meta-gradient with stop_gradient(g): dL/dw = sum_t 2 x_t * (2 t) = 60 > 0,
    # dL/db = sum_t 2 x_t * t = 30 > 0  ==> w = b = -0.01 after one Adam step.
    dLdw = sum(2 * 1.0 * 2 * t for t in range(6))
    dLdb = sum(2 * 1.0 * t for t in range(6))
    assert dLdw == 60 and dLdb == 30
    # unroll 2 with w = b = -0.01 starting from x = 1 (x carried by `update`)
    cfg = O.NetConfig("cw", (), "identity", None, 1.0, False)
    params = {"linear": {"w": np.full((1, 1), -0.01, np.float32),
                         "b": np.full((1,), -0.01, np.float32)}}
    r2 = O.unroll(O.Simple(), cfg, params, r1.x, (), 5)
    assert abs(float(r2.fx[-1]) - 0.7325327) < 5e-5          # places=4
    assert abs(float(r2.x) - 0.8559) < 5e-5
    assert abs(float(r2.x) - 0.8558813) < 1e-6


# ---------------------------------------------------------------- problems KATs
@pytest.mark.parametrize("value", [-1, 0, 1, 10])
def test_simple_values(value):
    assert O.Simple().f(np.float32(value)) == value ** 2
    assert O.SimpleMulti(1).f(np.array([value], np.float32)) == value ** 2


@pytest.mark.parametrize("value", [-1, 0, 1, 10])
def test_quadratic_values(value):
    w, y = 2.0, 3.0
    p = O.Quadratic(np.array([[[w]]], np.float32), np.array([[y]], np.float32))
    out = p.f(np.array([[value]], np.float32))
    assert out == ((w * value) - y) ** 2


def test_quadratic_shapes():
    rng = np.random.default_rng(0)
    p, x = O.Quadratic.sample(rng, batch_size=5, num_dims=3)
    assert x.shape == (5, 3) and p.w.shape == (5, 3, 3) and p.y.shape == (5, 3)
    assert np.ndim(p.f(x)) == 0 and p.f(x).dtype == np.float32
    assert p.grad(x).shape == x.shape and p.grad(x).dtype == np.float32


# ---------------------------------------------------------------- preprocess KATs
def test_clamp():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(100).astype(np.float32)
    assert np.all(O.clamp(x, min_value=0.0) >= 0)
    assert np.all(O.clamp(x, max_value=0.0) <= 0)
    assert np.all(O.clamp(x, min_value=0.0, max_value=0.0) == 0)
    assert O.clamp(x.reshape(50, 2), -1.0, 1.0).shape == (50, 2)


def test_log_and_sign():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 3)).astype(np.float32)
    assert O.log_and_sign(x, k=1).shape == (2, 6)
    out = O.log_and_sign(np.ones((1,), np.float32), k=10)
    assert abs(float(out[0])) < 1e-7                          # log part of 1.0 is 0
    x = rng.standard_normal((2, 1)).astype(np.float32)
    out = O.log_and_sign(x, k=1)
    assert np.all(np.sign(out[:, 1:]) == np.sign(x))
    # formula spot values (DM/preprocess.py:63-70), k = 5
    g = np.array([[1e-8], [-0.5], [3.0], [0.0]], np.float32)
    out = O.log_and_sign(g, k=5)
    eps = np.finfo(np.float32).eps
    exp_log = np.maximum(np.log(np.abs(g.astype(np.float64)) + eps) / 5, -1)
    exp_sign = np.clip(g.astype(np.float64) * np.exp(5), -1, 1)
    np.testing.assert_allclose(out[:, :1], exp_log, rtol=2e-6)
    np.testing.assert_allclose(out[:, 1:], exp_sign, rtol=2e-6)
    assert out.dtype == np.float32


# ---------------------------------------------------------------- networks KATs
@pytest.mark.parametrize("layers", [(), (1,), (20, 20)])
def test_zero_net_gives_exactly_zero_update(layers):
    cfg = O.NetConfig("cw", layers, "identity", None, 1.0, False)
    params = O.init_net_params(cfg, np.random.default_rng(0), initializer="zeros")
    g = np.random.default_rng(3).standard_normal((13, 7)).astype(np.float32)
    state = O.net_initial_state(cfg, g.size)
    delta, st = O.net_apply(cfg, params, g, state)
    assert delta.shape == g.shape
    assert np.all(delta == 0)


def test_net_variable_inventory():
    cfg = O.NetConfig("cw", (1,), "identity", None, 1.0, False)
    p = O.init_net_params(cfg, np.random.default_rng(0))
    # networks_test.py:40-49: 4 trainables for layers=(1,)
    assert sum(len(v) for v in p.values()) == 4
    assert p["lstm_1"]["w_gates"].shape == (2, 4) and p["linear"]["w"].shape == (1, 1)
    p = O.init_net_params(O.RNNPROP, np.random.default_rng(0))
    assert p["input_projection"]["w"].shape == (2, 20)
    assert p["lstm_1"]["w_gates"].shape == (40, 80)
    assert p["lstm_2"]["w_gates"].shape == (40, 80)
    p = O.init_net_params(O.DM_LOGSIGN, np.random.default_rng(0))
    assert p["lstm_1"]["w_gates"].shape == (22, 80)


def test_sgd_and_adam_nets():
    g = np.random.default_rng(4).standard_normal((10, 3)).astype(np.float32)
    np.testing.assert_array_equal(O.sgd_net(g, 0.25), -np.float32(0.25) * g)
    st = (np.float32(0), np.zeros((30, 1), np.float32), np.zeros((30, 1), np.float32))
    upd, st2 = O.adam_net(g, st, learning_rate=0.0)
    assert upd.shape == g.shape and np.all(upd == 0)
    upd, _ = O.adam_net(g, st, learning_rate=1e-3)
    # first Adam step == -lr * sign(g) up to epsilon
    np.testing.assert_allclose(upd, -1e-3 * np.sign(g), rtol=1e-4)


# ------------------------------------------------ independent cross-checks
def _torch_lstm_from_sonnet(w_gates, b_gates, in_dim, H):
    """Sonnet (i, j, f, o) + forget_bias 1  ->  torch (i, f, g, o)."""
    cell = torch.nn.LSTMCell(in_dim, H).double()
    perm = np.concatenate([np.arange(0, H), np.arange(2 * H, 3 * H),
                           np.arange(H, 2 * H), np.arange(3 * H, 4 * H)])
    w = w_gates.astype(np.float64)[:, perm]
    b = b_gates.astype(np.float64)[perm].copy()
    b[H:2 * H] += 1.0
    with torch.no_grad():
        cell.weight_ih.copy_(torch.from_numpy(w[:in_dim].T.copy()))
        cell.weight_hh.copy_(torch.from_numpy(w[in_dim:].T.copy()))
        cell.bias_ih.copy_(torch.from_numpy(b))
        cell.bias_hh.zero_()
    return cell


@pytest.mark.parametrize("in_dim,H", [(1, 20), (2, 20), (20, 20), (3, 5)])
def test_lstm_cell_matches_torch_lstmcell(in_dim, H):
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((in_dim + H, 4 * H)) * 0.4).astype(np.float32)
    b = (rng.standard_normal((4 * H,)) * 0.3).astype(np.float32)
    x = rng.standard_normal((37, in_dim)).astype(np.float32)
    h = rng.standard_normal((37, H)).astype(np.float32) * 0.5
    c = rng.standard_normal((37, H)).astype(np.float32)
    h2, c2 = O.lstm_cell(x, h, c, w, b)
    assert h2.dtype == np.float32 and c2.dtype == np.float32
    cell = _torch_lstm_from_sonnet(w, b, in_dim, H)
    with torch.no_grad():
        th, tc = cell(torch.from_numpy(x).double(),
                      (torch.from_numpy(h).double(), torch.from_numpy(c).double()))
    np.testing.assert_allclose(h2, th.numpy(), atol=1e-6)
    np.testing.assert_allclose(c2, tc.numpy(), atol=2e-6)


def _torch_losses():
    """The reference's loss formulas restated in torch (autograd gives the
    gradient the way tf.gradients would)."""
    def quadratic(x, p):
        prod = torch.matmul(p["w"], x.unsqueeze(-1)).squeeze(-1)      # problems.py:98
        return torch.mean(torch.sum((prod - p["y"]) ** 2, 1))          # :99

    def lasso(x, p):
        prod = torch.matmul(p["w"], x.unsqueeze(-1))                    # :128
        left = 0.5 * torch.sum((prod - p["y"]) ** 2, 1)                 # :129
        other = p["l"] * torch.sum(torch.abs(x), dim=1, keepdim=True)   # :130
        return torch.mean(left + other)                                 # :131

    def rastrigin(x, p):
        prod = torch.matmul(p["A"], x)                                  # :206
        ras = torch.linalg.norm((prod - p["B"]).flatten(1), dim=1)      # :207
        cq = torch.matmul(p["C"].transpose(1, 2), torch.cos(2 * np.pi * x)).squeeze()  # :209
        return torch.mean(0.5 * ras ** 2 - p["alpha"] * cq + p["alpha"] * x.shape[1])  # :211

    def square_cos(x, p):
        prod = torch.matmul(p["w"], x.unsqueeze(-1)).squeeze(-1)
        prod2 = torch.matmul(p["wcos"], (10 * torch.cos(2 * 3.1415926 * x)).unsqueeze(-1)).squeeze(-1)
        prod3 = torch.sum((prod - p["y"]) ** 2, 1) - torch.sum(prod2, 1) + 10 * x.shape[1]
        return torch.mean(prod3)
    return quadratic, lasso, rastrigin, square_cos


def _t(a):
    return torch.from_numpy(np.asarray(a, np.float64))


@pytest.mark.parametrize("B,D", [(4, 3), (7, 10), (2, 16)])
def test_problem_values_and_grads_match_torch_autograd(B, D):
    rng = np.random.default_rng(6)
    tq, tl, tr, ts = _torch_losses()

    prob, x = O.Quadratic.sample(rng, B, D, stddev=0.5, dtype=np.float64)
    xt = _t(x).requires_grad_(True)
    f = tq(xt, {"w": _t(prob.w), "y": _t(prob.y)})
    f.backward()
    np.testing.assert_allclose(prob.f(x), f.item(), rtol=1e-12)
    np.testing.assert_allclose(prob.grad(x), xt.grad.numpy(), rtol=1e-10, atol=1e-12)

    prob, x = O.Lasso.sample(rng, B, D, stddev=0.5, l=0.1, num_rows=D + 2, dtype=np.float64)
    xt = _t(x).requires_grad_(True)
    f = tl(xt, {"w": _t(prob.w), "y": _t(prob.y), "l": 0.1})
    f.backward()
    np.testing.assert_allclose(prob.f(x), f.item(), rtol=1e-12)
    np.testing.assert_allclose(prob.grad(x), xt.grad.numpy(), rtol=1e-10, atol=1e-12)

    prob, x = O.Rastrigin.sample(rng, B, D, dtype=np.float64)
    xt = _t(x).requires_grad_(True)
    f = tr(xt, {"A": _t(prob.A), "B": _t(prob.B), "C": _t(prob.C), "alpha": 10})
    f.backward()
    np.testing.assert_allclose(prob.f(x), f.item(), rtol=1e-12)
    np.testing.assert_allclose(prob.grad(x), xt.grad.numpy(), rtol=1e-9, atol=1e-11)

    prob, x = O.SquareCos.sample(rng, B, D, stddev=0.5, dtype=np.float64)
    xt = _t(x).requires_grad_(True)
    f = ts(xt, {"w": _t(prob.w), "y": _t(prob.y), "wcos": _t(prob.wcos)})
    f.backward()
    np.testing.assert_allclose(prob.f(x), f.item(), rtol=1e-12)
    np.testing.assert_allclose(prob.grad(x), xt.grad.numpy(), rtol=1e-9, atol=1e-11)


def test_sharded_batch_keeps_global_mean():
    """with B sharded, 1/B stays the global B."""
    rng = np.random.default_rng(7)
    prob, x = O.Quadratic.sample(rng, 8, 5, stddev=0.3)
    g_full = prob.grad(x)
    f_full = prob.f(x)
    fs, gs = 0.0, []
    for lo in (0, 4):
        sh = O.Quadratic(prob.w[lo:lo + 4], prob.y[lo:lo + 4], batch_global=8)
        fs += sh.f(x[lo:lo + 4])
        gs.append(sh.grad(x[lo:lo + 4]))
    np.testing.assert_allclose(fs, f_full, rtol=1e-6)
    np.testing.assert_allclose(np.concatenate(gs), g_full, rtol=1e-6)


def test_unroll_whole_net_matches_torch_float64():
    """Whole-step cross-check of net_apply + unroll (LogAndSign and RNNProp
    wiring) against an independent torch float64 restatement."""
    rng = np.random.default_rng(8)
    B, D, T = 3, 4, 6
    for cfg in (O.DM_IDENTITY, O.DM_LOGSIGN, O.RNNPROP):
        params = O.init_net_params(cfg, rng, dtype=np.float64)
        prob, x0 = O.Quadratic.sample(rng, B, D, stddev=0.5, dtype=np.float64)
        res = O.unroll(prob, cfg, params, x0, O.net_initial_state(cfg, B * D, np.float64), T,
                       step0=1)
        # torch restatement
        tq = _torch_losses()[0]
        w, y = _t(prob.w), _t(prob.y)
        x = _t(x0)
        H = 20
        cells = []
        in_dim = cfg.in_dim
        for li in (1, 2):
            p = params["lstm_%d" % li]
            cells.append(_torch_lstm_from_sonnet(p["w_gates"], p["b_gates"], in_dim, H))
            in_dim = H
        st = [(torch.zeros(B * D, H, dtype=torch.float64),) * 2 for _ in range(2)]
        m = torch.zeros_like(x)
        v = torch.zeros_like(x)
        fx = []
        for t in range(T):
            xr = x.clone().requires_grad_(True)
            f = tq(xr, {"w": w, "y": y})
            f.backward()
            fx.append(f.item())
            g = xr.grad
            if cfg.kind == "rnnprop":
                k = float(1 + t)
                m = 0.95 * m + 0.05 * g
                v = 0.95 * v + 0.05 * g * g
                mh = m / (1 - 0.95 ** k)
                vh = v / (1 - 0.95 ** k)
                feats = torch.stack([(mh / (vh.sqrt() + 1e-8)).reshape(-1),
                                     (g / (vh.sqrt() + 1e-8)).reshape(-1)], -1)
                ip = params["input_projection"]
                feats = torch.nn.functional.elu(feats @ _t(ip["w"]) + _t(ip["b"]))
            elif cfg.preprocess_name == "LogAndSign":
                gf = g.reshape(-1, 1)
                eps = float(np.finfo(np.float64).eps)
                feats = torch.cat([torch.clamp(torch.log(gf.abs() + eps) / 5, min=-1.0),
                                   torch.clamp(gf * float(np.exp(5)), -1.0, 1.0)], 1)
            else:
                feats = g.reshape(-1, 1)
            out = feats
            with torch.no_grad():
                for li in range(2):
                    st[li] = cells[li](out, st[li])
                    out = st[li][0]
                lin = params["linear"]
                d = out @ _t(lin["w"]) + _t(lin["b"])
                d = (torch.tanh(d) if cfg.tanh_output else d) * cfg.scale
                x = x + d.reshape(x.shape)
        fx.append(tq(x, {"w": w, "y": y}).item())
        np.testing.assert_allclose(res.fx, np.array(fx), rtol=1e-9)
        np.testing.assert_allclose(res.x, x.numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(res.state[1][0], st[1][0].numpy(), rtol=1e-8, atol=1e-12)
