"""NumPy fp32 restatement of the Open-L2O model-free inner unroll loop.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Every function cites the
reference file:line it restates.  ``DM/`` is shorthand for
``/root/reference/Model_Free_L2O/L2O-DM and L2O-RNNProp/``.

All arithmetic is done in float32 (like the reference's ``dtype=tf.float32``);
pass ``dtype=np.float64`` to the entry points to obtain a high-precision
trajectory that the tests use to bound how far two fp32 implementations may
legitimately drift apart.
"""
from __future__ import annotations

import collections

import numpy as np

__all__ = [
    "clamp", "log_and_sign", "sigmoid", "elu", "lstm_cell", "linear",
    "NetConfig", "init_net_params", "net_initial_state", "net_apply",
    "Simple", "SimpleMulti", "Quadratic", "Lasso", "Rastrigin", "SquareCos",
    "unroll", "UnrollResult", "sgd_net", "adam_net", "truncated_normal", "MnistMLP", "unroll_multi", "rnnprop_inputs",
    "net_bwd_step", "preprocess_bwd", "tf_adam_step",
    "DM_IDENTITY", "DM_LOGSIGN", "RNNPROP",
]


# ----------------------------------------------------------------------------
# preprocess  (DM/preprocess.py)
# ----------------------------------------------------------------------------
def clamp(x, min_value=None, max_value=None):
    """DM/preprocess.py:26-39 (Clamp._build)."""
    out = x
    if min_value is not None:
        out = np.maximum(out, x.dtype.type(min_value))
    if max_value is not None:
        out = np.minimum(out, x.dtype.type(max_value))
    return out


def log_and_sign(g, k):
    """DM/preprocess.py:52-70 (LogAndSign._build).

    ``g`` has shape [..., d]; the result has shape [..., 2d] with the clamped
    log first and the clamped scaled sign second.  ``eps`` is the machine
    epsilon of the gradient dtype (:63); ``exp(k)`` is a python double that TF
    converts to the tensor dtype before the multiply (:68).
    """
    dt = g.dtype.type
    eps = np.finfo(g.dtype).eps
    log = np.log(np.abs(g) + dt(eps))
    clamped_log = clamp(log / dt(k), min_value=-1.0)
    sign = clamp(g * dt(np.exp(k)), min_value=-1.0, max_value=1.0)
    return np.concatenate([clamped_log, sign], axis=g.ndim - 1)


# ----------------------------------------------------------------------------
# dm-sonnet 1.11 pieces (NOT in the reference tree; restated from Sonnet v1's
# published gated_rnn.LSTM / basic.Linear / DeepRNN semantics)
# ----------------------------------------------------------------------------
def sigmoid(x):
    one = x.dtype.type(1.0)
    with np.errstate(over="ignore"):          # exp(+large) -> inf -> sigmoid 0, as in TF
        return one / (one + np.exp(-x))


def elu(x):
    """tf.nn.elu used at DM/networks.py:219."""
    return np.where(x > 0, x, np.expm1(np.minimum(x, x.dtype.type(0))))


def linear(x, w, b):
    """snt.Linear: y = x @ w + b (call sites DM/networks.py:183, 203)."""
    return x @ w + b


def lstm_cell(x, h, c, w_gates, b_gates, forget_bias=1.0):
    """snt.LSTM._build (dm-sonnet 1.11; call site DM/networks.py:197).

    gates = [x, h] @ w_gates + b_gates ; i, j, f, o = split(gates, 4)
    c' = sigmoid(f + forget_bias) * c + sigmoid(i) * tanh(j)
    h' = tanh(c') * sigmoid(o)
    """
    z = np.concatenate([x, h], axis=1) @ w_gates + b_gates
    H = h.shape[1]
    i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
    c_next = sigmoid(f + z.dtype.type(forget_bias)) * c + sigmoid(i) * np.tanh(j)
    h_next = np.tanh(c_next) * sigmoid(o)
    return h_next, c_next


def truncated_normal(rng, shape, stddev, dtype=np.float32):
    """tf.truncated_normal_initializer: resample anything beyond 2 sigma."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(dtype)


# ----------------------------------------------------------------------------
# optimizer networks  (DM/networks.py)
# ----------------------------------------------------------------------------
class NetConfig(collections.namedtuple(
        "NetConfig", "kind layers preprocess_name preprocess_options scale tanh_output")):
    """Options of StandardDeepLSTM.__init__ (DM/networks.py:157-205).

    kind: "cw" (CoordinateWiseDeepLSTM, :239) or "rnnprop" (RNNprop, :279).
    """

    @property
    def in_dim(self):
        if self.kind == "rnnprop":
            if self.preprocess_name == "fc":
                return int(self.preprocess_options["dim"])
            return 2
        if self.preprocess_name == "LogAndSign":
            return 2
        return 1


# the three harness configurations, DM/util.py:136-143 / :99-109 / :251-263
DM_IDENTITY = NetConfig("cw", (20, 20), "identity", None, 1.0, False)
DM_LOGSIGN = NetConfig("cw", (20, 20), "LogAndSign", {"k": 5}, 0.01, False)
RNNPROP = NetConfig("rnnprop", (20, 20), "fc", {"dim": 20}, 0.01, True)


def init_net_params(cfg, rng, dtype=np.float32, initializer=None):
    """Sonnet default initialisers in the ``.l2l`` dict layout of
    networks.save (DM/networks.py:47-62): {module: {variable: ndarray}}.

    snt.Linear: w ~ TruncNormal(1/sqrt(in)), b = 0.  snt.LSTM: w_gates
    [in+H, 4H] ~ TruncNormal(1/sqrt(in+H)), b_gates = 0.  ``initializer=
    "zeros"`` reproduces networks_test.py's all-zero nets.
    """
    def mk(shape, fan_in):
        if initializer == "zeros":
            return np.zeros(shape, dtype)
        return truncated_normal(rng, shape, 1.0 / np.sqrt(fan_in), dtype)

    params = {}
    raw_in = 2 if cfg.kind == "rnnprop" else 1
    if cfg.preprocess_name == "fc":
        dim = int(cfg.preprocess_options["dim"])
        params["input_projection"] = {"w": mk((raw_in, dim), raw_in),
                                      "b": np.zeros((dim,), dtype)}
    size_in = cfg.in_dim
    for li, H in enumerate(cfg.layers, start=1):
        params["lstm_%d" % li] = {"w_gates": mk((size_in + H, 4 * H), size_in + H),
                                  "b_gates": np.zeros((4 * H,), dtype)}
        size_in = H
    params["linear"] = {"w": mk((size_in, 1), size_in), "b": np.zeros((1,), dtype)}
    return params


def net_initial_state(cfg, n, dtype=np.float32):
    """DeepRNN.initial_state: per layer (hidden, cell) zeros [n, H]
    (DM/networks.py:234-236, 273-276)."""
    return tuple((np.zeros((n, H), dtype), np.zeros((n, H), dtype)) for H in cfg.layers)


def net_apply(cfg, params, inputs, state):
    """StandardDeepLSTM._build (DM/networks.py:207-232) behind
    CoordinateWiseDeepLSTM._build (:254-271) / RNNprop._build (:287-295).

    inputs: cw -> gradient of any shape; rnnprop -> tuple (m_tilde, g_tilde).
    Returns (delta shaped like the gradient, next_state).
    """
    if cfg.kind == "rnnprop":
        m, g = inputs
        out_shape = g.shape
        feats = np.stack([m.reshape(-1), g.reshape(-1)], axis=-1)      # :289-290
    else:
        out_shape = inputs.shape
        feats = inputs.reshape(-1, 1)                                   # :251-252
    dt = feats.dtype.type
    if cfg.preprocess_name == "fc":                                     # :218-219
        p = params["input_projection"]
        feats = elu(linear(feats, p["w"], p["b"]))
    elif cfg.preprocess_name == "LogAndSign":                           # :221
        feats = log_and_sign(feats[..., None], **cfg.preprocess_options)
    elif cfg.preprocess_name == "identity":
        feats = feats[..., None]
    else:
        raise ValueError(cfg.preprocess_name)
    feats = feats.reshape(feats.shape[0], -1)                           # :224
    next_state = []
    out = feats
    for li, (h, c) in enumerate(state, start=1):                        # :225 DeepRNN
        p = params["lstm_%d" % li]
        h2, c2 = lstm_cell(out, h, c, p["w_gates"], p["b_gates"])
        next_state.append((h2, c2))
        out = h2
    p = params["linear"]
    final = linear(out, p["w"], p["b"])                                 # :227
    if cfg.tanh_output:                                                 # :229-232
        final = np.tanh(final) * dt(cfg.scale)
    else:
        final = final * dt(cfg.scale)
    return final.reshape(out_shape), tuple(next_state)


def sgd_net(g, learning_rate=0.001):
    """networks.Sgd._build, DM/networks.py:367-368."""
    return -g.dtype.type(learning_rate) * g


def adam_net(g, state, learning_rate=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """networks.Adam._build, DM/networks.py:394-412.  state = (t, m, v)."""
    dt = g.dtype.type
    t, m, v = state
    t_next = t + 1
    gf = g.reshape(-1, 1)
    m_next = dt(beta1) * m + dt(1 - beta1) * gf
    m_hat = m_next / dt(1 - np.power(dt(beta1), dt(t_next)))
    v_next = dt(beta2) * v + dt(1 - beta2) * np.square(gf)
    v_hat = v_next / dt(1 - np.power(dt(beta2), dt(t_next)))
    upd = -dt(learning_rate) * m_hat / (np.sqrt(v_hat) + dt(epsilon))
    return upd.reshape(g.shape), (t_next, m_next, v_next)


# ----------------------------------------------------------------------------
# optimizee problems  (DM/problems.py).  ``batch_global`` is the batch size in
# the reduce_mean: when a shard of the batch is evaluated the 1/B factor must
# stay the GLOBAL one.
# ----------------------------------------------------------------------------
class _Problem(object):
    batch_global = None

    def _bg(self, b):
        return b if self.batch_global is None else self.batch_global


class Simple(_Problem):
    """problems.simple, DM/problems.py:41-53: f(x) = x^2, x0 = 1."""

    def init_x(self, rng=None, dtype=np.float32):
        return np.ones((), dtype)

    def f(self, x):
        return np.square(x)

    def grad(self, x):
        return x.dtype.type(2) * x


class SimpleMulti(_Problem):
    """problems.simple_multi_optimizer, DM/problems.py:56-70 (flattened to one
    vector of num_dims scalars)."""

    def __init__(self, num_dims=2):
        self.num_dims = num_dims

    def init_x(self, rng=None, dtype=np.float32):
        return np.ones((self.num_dims,), dtype)

    def f(self, x):
        return np.sum(np.square(x))

    def grad(self, x):
        return x.dtype.type(2) * x


class Quadratic(_Problem):
    """problems.quadratic, DM/problems.py:73-101.

    x [B,D] ~ N(0, stddev^2); w [B,D,D], y [B,D] ~ U[0,1).
    f = mean_b sum_i (w_b x_b - y_b)_i^2      (:98-99)
    """

    def __init__(self, w, y, batch_global=None):
        self.w, self.y, self.batch_global = w, y, batch_global

    @staticmethod
    def sample(rng, batch_size=128, num_dims=10, stddev=0.01, dtype=np.float32):
        w = rng.random((batch_size, num_dims, num_dims)).astype(dtype)
        y = rng.random((batch_size, num_dims)).astype(dtype)
        x = (rng.standard_normal((batch_size, num_dims)) * stddev).astype(dtype)
        return Quadratic(w, y), x

    def residual(self, x):
        return np.matmul(self.w, x[..., None])[..., 0] - self.y

    def f_per_problem(self, x):
        r = self.residual(x)
        return np.sum(r * r, axis=1)

    def f(self, x):
        return np.sum(self.f_per_problem(x)) / x.dtype.type(self._bg(x.shape[0]))

    def grad(self, x):
        r = self.residual(x)
        g = np.matmul(np.swapaxes(self.w, 1, 2), r[..., None])[..., 0]
        return g * x.dtype.type(2.0 / self._bg(x.shape[0]))


class Lasso(_Problem):
    """problems.lasso (DM/problems.py:103-134) and lasso_fixed (:137-175).

    w [B,M,N] (square for ``lasso``), y [B,M,1], x [B,N];
    f = mean_b [ 0.5 ||w_b x_b - y_b||^2 + l ||x_b||_1 ]     (:128-131)
    d|x|/dx = sign(x) with sign(0) = 0 (TF's gradient of tf.abs).
    """

    def __init__(self, w, y, l=0.005, batch_global=None):
        self.w, self.y, self.l, self.batch_global = w, y, l, batch_global

    @staticmethod
    def sample(rng, batch_size=128, num_dims=10, stddev=0.01, l=0.005,
               num_rows=None, dtype=np.float32):
        m = num_dims if num_rows is None else num_rows
        w = rng.random((batch_size, m, num_dims)).astype(dtype)
        y = rng.random((batch_size, m, 1)).astype(dtype)
        x = (rng.standard_normal((batch_size, num_dims)) * stddev).astype(dtype)
        return Lasso(w, y, l), x

    def residual(self, x):
        return np.matmul(self.w, x[..., None]) - self.y          # [B,M,1]

    def f_per_problem(self, x):
        dt = x.dtype.type
        r = self.residual(x)
        left = dt(0.5) * np.sum(r * r, axis=1)[:, 0]
        other = dt(self.l) * np.sum(np.abs(x), axis=1)
        return left + other

    def f(self, x):
        return np.sum(self.f_per_problem(x)) / x.dtype.type(self._bg(x.shape[0]))

    def grad(self, x):
        dt = x.dtype.type
        r = self.residual(x)
        g = np.matmul(np.swapaxes(self.w, 1, 2), r)[..., 0] + dt(self.l) * np.sign(x)
        return g / dt(self._bg(x.shape[0]))


class Rastrigin(_Problem):
    """problems.rastrigin, DM/problems.py:177-213.

    x [B,D,1] ~ N(0, stddev^2); A [B,D,D], B, C [B,D,1] ~ N(0, stddev^2).
    f = mean_b [ 0.5 ||A x - B||^2 - alpha C^T cos(2 pi x) + alpha D ]   (:206-211)
    (TF computes the norm, then squares it; the gradient of 0.5*norm^2 is the
    residual itself.)
    """

    def __init__(self, A, B, C, alpha=10, batch_global=None):
        self.A, self.B, self.C, self.alpha, self.batch_global = A, B, C, alpha, batch_global

    @staticmethod
    def sample(rng, batch_size=128, num_dims=10, alpha=10, stddev=1, dtype=np.float32):
        x = (rng.standard_normal((batch_size, num_dims, 1)) * stddev).astype(dtype)
        A = (rng.standard_normal((batch_size, num_dims, num_dims)) * stddev).astype(dtype)
        B = (rng.standard_normal((batch_size, num_dims, 1)) * stddev).astype(dtype)
        C = (rng.standard_normal((batch_size, num_dims, 1)) * stddev).astype(dtype)
        return Rastrigin(A, B, C, alpha), x

    def f_per_problem(self, x):
        dt = x.dtype.type
        r = np.matmul(self.A, x) - self.B
        ras_norm = np.sqrt(np.sum(r * r, axis=(1, 2)))
        cq = np.sum(self.C * np.cos(dt(2 * np.pi) * x), axis=(1, 2))
        D = x.shape[1]
        return dt(0.5) * ras_norm ** 2 - dt(self.alpha) * cq + dt(self.alpha * D)

    def f(self, x):
        return np.sum(self.f_per_problem(x)) / x.dtype.type(self._bg(x.shape[0]))

    def grad(self, x):
        dt = x.dtype.type
        r = np.matmul(self.A, x) - self.B
        g = np.matmul(np.swapaxes(self.A, 1, 2), r)
        g = g + dt(2 * np.pi * self.alpha) * self.C * np.sin(dt(2 * np.pi) * x)
        return g / dt(self._bg(x.shape[0]))


class SquareCos(_Problem):
    """problems.square_cos, DM/problems.py:959-994.

    f = mean_b [ ||w x - y||^2 - sum_i (wcos (10 cos(2*3.1415926 x)))_i + 10 D ]
    """

    def __init__(self, w, y, wcos, batch_global=None):
        self.w, self.y, self.wcos, self.batch_global = w, y, wcos, batch_global

    @staticmethod
    def sample(rng, batch_size=128, num_dims=10, stddev=0.01, dtype=np.float32):
        w = rng.random((batch_size, num_dims, num_dims)).astype(dtype)
        y = rng.random((batch_size, num_dims)).astype(dtype)
        wcos = rng.random((batch_size, num_dims, num_dims)).astype(dtype)
        x = (rng.standard_normal((batch_size, num_dims)) * stddev).astype(dtype)
        return SquareCos(w, y, wcos), x

    def f_per_problem(self, x):
        dt = x.dtype.type
        r = np.matmul(self.w, x[..., None])[..., 0] - self.y
        c = dt(10) * np.cos(dt(2 * 3.1415926) * x)
        p2 = np.matmul(self.wcos, c[..., None])[..., 0]
        return np.sum(r * r, axis=1) - np.sum(p2, axis=1) + dt(10 * x.shape[1])

    def f(self, x):
        return np.sum(self.f_per_problem(x)) / x.dtype.type(self._bg(x.shape[0]))

    def grad(self, x):
        dt = x.dtype.type
        r = np.matmul(self.w, x[..., None])[..., 0] - self.y
        g = dt(2) * np.matmul(np.swapaxes(self.w, 1, 2), r[..., None])[..., 0]
        colsum = np.sum(self.wcos, axis=1)                       # d/dc_j sum_i (wcos c)_i
        g = g + colsum * dt(10) * dt(2 * 3.1415926) * np.sin(dt(2 * 3.1415926) * x)
        return g / dt(self._bg(x.shape[0]))


class MnistMLP(_Problem):
    """problems.mnist, DM/problems.py:246-288: snt.nets.MLP([hidden, 10]) on flattened images,
    loss = mean sparse-softmax cross-entropy of a minibatch gathered with ``indices``.
    Variables (Sonnet names): mlp/linear_0/w [n_in,H], mlp/linear_0/b [H], mlp/linear_1/w [H,O],
    mlp/linear_1/b [O], all ~ N(0, 0.01^2) (_nn_initializers, :35-38)."""

    def __init__(self, images, labels, activation="sigmoid"):
        self.images = images.reshape(images.shape[0], -1)
        self.labels = labels
        self.activation = activation

    def init_vars(self, rng, hidden=20, n_out=10, dtype=np.float32):
        n_in = self.images.shape[1]
        widths = [n_in] + ([hidden] if np.isscalar(hidden) else list(hidden)) + [n_out]
        return [(rng.standard_normal(shape) * 0.01).astype(dtype)
                for l in range(len(widths) - 1) for shape in ((widths[l], widths[l + 1]), (widths[l + 1],))]

    def fg_deep(self, variables, indices, want_grad=True):
        """Any number of hidden layers (snt.nets.MLP(list(layers) + [10]), DM/problems.py:275-276; "mnist_deeper" =
        layers (20, 20), DM/util.py:157-163): variables = [w0, b0, w1, b1, ..., wL, bL]."""
        ws, bs = variables[0::2], variables[1::2]
        dt = ws[0].dtype.type
        x = self.images[indices].astype(ws[0].dtype)
        lab = self.labels[indices]
        acts = [x]
        for l in range(len(ws) - 1):
            a = acts[-1] @ ws[l] + bs[l]
            acts.append(sigmoid(a) if self.activation == "sigmoid" else np.maximum(a, dt(0)))
        z = acts[-1] @ ws[-1] + bs[-1]
        zmax = z.max(axis=1, keepdims=True)
        lse = zmax[:, 0] + np.log(np.sum(np.exp(z - zmax), axis=1))
        n = np.arange(len(indices))
        loss = np.sum(lse - z[n, lab]) / dt(len(indices))
        if not want_grad:
            return loss, None
        d = np.exp(z - lse[:, None])
        d[n, lab] -= dt(1)
        d = d / dt(len(indices))
        grads = [None] * len(variables)
        for l in range(len(ws) - 1, -1, -1):
            grads[2 * l] = acts[l].T @ d
            grads[2 * l + 1] = d.sum(axis=0)
            if l > 0:
                h = acts[l]
                d = d @ ws[l].T
                d = d * h * (dt(1) - h) if self.activation == "sigmoid" else d * (h > 0)
        return loss, grads

    def fg(self, variables, indices, want_grad=True):
        if len(variables) != 4:
            return self.fg_deep(variables, indices, want_grad)
        w1, b1, w2, b2 = variables
        dt = w1.dtype.type
        x = self.images[indices].astype(w1.dtype)
        lab = self.labels[indices]
        a = x @ w1 + b1
        h = sigmoid(a) if self.activation == "sigmoid" else np.maximum(a, dt(0))
        z = h @ w2 + b2
        zmax = z.max(axis=1, keepdims=True)
        lse = zmax[:, 0] + np.log(np.sum(np.exp(z - zmax), axis=1))
        n = np.arange(len(indices))
        loss = np.sum(lse - z[n, lab]) / dt(len(indices))
        if not want_grad:
            return loss, None
        dz = np.exp(z - lse[:, None])
        dz[n, lab] -= dt(1)
        dz = dz / dt(len(indices))
        gw2 = h.T @ dz
        gb2 = dz.sum(axis=0)
        dh = dz @ w2.T
        dh = dh * h * (dt(1) - h) if self.activation == "sigmoid" else dh * (h > 0)
        gw1 = x.T @ dh
        gb1 = dh.sum(axis=0)
        return loss, [gw1, gb1, gw2, gb2]


def rnnprop_inputs(g, m, v, k, beta1=0.95, beta2=0.95):
    """RNNProp's network inputs, DM/meta_rnnprop_train.py:383-388 (== DM/meta_rnnprop_eval.py `update`):
    m' = b1 m + (1 - b1) g ; v' = b2 v + (1 - b2) g^2 (carried un-debiased); m^ = m'/(1 - b1^k), v^ = v'/(1 - b2^k);
    (m~, g~) = (m^, g) / (sqrt(v^) + 1e-8), k = the fed `step` + t.  Returns ((m~, g~), m', v')."""
    dt = g.dtype.type
    k = dt(k)
    m = dt(beta1) * m + dt(1.0 - beta1) * g
    m_hat = m / (dt(1) - np.power(dt(beta1), k))
    v = dt(beta2) * v + dt(1.0 - beta2) * g * g
    v_hat = v / (dt(1) - np.power(dt(beta2), k))
    den = np.sqrt(v_hat) + dt(1e-8)
    return (m_hat / den, g / den), m, v


def unroll_multi(fg, cfg, params, variables, states, T, ms=None, vs=None, step0=1, beta1=0.95, beta2=0.95,
                 return_moments=False):
    """The unroll of DM/meta.py:338-376 for an optimizee with SEVERAL variables that share one
    coordinate-wise net (default net_assignments): ``fg(variables, t, want_grad) ->
    (loss, [grad per variable])``; states = one net state per variable.  RNNProp nets
    (cfg.kind == "rnnprop", DM/meta_rnnprop_train.py:371-423): one pair of moments per variable (``ms`` / ``vs``,
    zeros by default), inputs of ``rnnprop_inputs`` with exponent step0 + t."""
    variables = [v.copy() for v in variables]
    states = list(states)
    rn = cfg.kind == "rnnprop"
    if rn:
        ms = [np.zeros_like(v) for v in variables] if ms is None else [m.copy() for m in ms]
        vs = [np.zeros_like(v) for v in variables] if vs is None else [v.copy() for v in vs]
    fx = np.zeros((T + 1,), variables[0].dtype)
    for t in range(T):
        fx[t], grads = fg(variables, t, True)
        for j, g in enumerate(grads):
            if rn:
                inputs, ms[j], vs[j] = rnnprop_inputs(g, ms[j], vs[j], step0 + t, beta1, beta2)
                delta, states[j] = net_apply(cfg, params, inputs, states[j])
            else:
                delta, states[j] = net_apply(cfg, params, g, states[j])
            variables[j] = variables[j] + delta
    fx[T], _ = fg(variables, T, False)
    if return_moments:
        return fx, variables, states, ms, vs
    return fx, variables, states


# ----------------------------------------------------------------------------
# the unroll  (DM/meta.py:319-389; scaled variant DM/meta_dm_train.py:378-419;
# RNNProp DM/meta_rnnprop_train.py:371-423 == DM/meta_rnnprop_eval.py)
# ----------------------------------------------------------------------------
UnrollResult = collections.namedtuple("UnrollResult", "fx x state m v loss")


def unroll(problem, cfg, params, x0, state0, T, x_scale=None,
           m0=None, v0=None, step0=1, beta1=0.95, beta2=0.95):
    """One ``sess.run`` of MetaLoss: T optimizer steps + the final evaluation.

    for t in 0..T-1:   fx[t] = f(x_t * s)                     meta.py:344-345
                       g = s * grad f(x_t * s)   (constant)   meta.py:322-329
                       delta, state = net(g, state)           meta.py:332
                       x_{t+1} = x_t + delta                  meta.py:353
    fx[T] = f(x_T * s) ; loss = sum(fx)                       meta.py:372-376

    RNNProp (cfg.kind == "rnnprop") feeds Adam-normalised inputs,
    meta_rnnprop_train.py:383-388 with exponent ``step0 + t`` (step0 is the
    harness-fed ``step`` placeholder, util.py:59-60).

    Returns fx[0..T], x_T, state_T, (m_T, v_T), loss.  The harness' ``update``
    op (meta.py:387-389) is "carry x_T/state_T/m_T/v_T into the next call".
    """
    dt = x0.dtype.type
    x = x0.copy()
    state = state0
    s = None if x_scale is None else x_scale.astype(x0.dtype)
    m = None if m0 is None else m0.copy()
    v = None if v0 is None else v0.copy()
    if cfg.kind == "rnnprop" and m is None:
        m = np.zeros_like(x0)
        v = np.zeros_like(x0)
    fx = np.zeros((T + 1,), x0.dtype)
    for t in range(T):
        xs = x if s is None else x * s
        fx[t] = problem.f(xs)
        g = problem.grad(xs)
        if s is not None:
            g = g * s
        if cfg.kind == "rnnprop":
            inputs, m, v = rnnprop_inputs(g, m, v, step0 + t, beta1, beta2)
            delta, state = net_apply(cfg, params, inputs, state)
        else:
            delta, state = net_apply(cfg, params, g, state)
        x = x + delta
    xs = x if s is None else x * s
    fx[T] = problem.f(xs)
    return UnrollResult(fx, x, state, m, v, np.sum(fx))


# ----------------------------------------------------------------------------
# meta-gradient: what tf.train.AdamOptimizer(lr).minimize(loss) differentiates in
# MetaOptimizer.meta_minimize (DM/meta.py:398-414) with g = stop_gradient(grad f)
# (DM/meta.py:328-329).  One BPTT step of the optimizer network, hand-derived; the tests pin
# it against torch autograd of the restated unroll.
# ----------------------------------------------------------------------------
def net_bwd_step(cfg, params, inputs, state_prev, dx_next, carry_in):
    """inputs: g (cw) or (m_tilde, g_tilde) (rnnprop), flat [N].  state_prev: the net state BEFORE
    the step.  dx_next [N] = dL/dx_{t+1}.  carry_in = (dh1, dc1, dh2, dc2) [N,H] from step t+1.
    Returns (carry_out, rows) with rows = dict(act1, dz1, act2, dz2, h2, dd[, feats, du]) such that
    dW1 = act1^T dz1, db1 = sum dz1, dW2 = act2^T dz2, db2 = sum dz2, dw_lin = h2^T dd, db_lin = sum dd,
    dW_fc = feats^T du, db_fc = sum du."""
    dt = dx_next.dtype.type
    rows = {}
    if cfg.kind == "rnnprop":
        m, g = inputs
        feats = np.stack([m.reshape(-1), g.reshape(-1)], -1)
        p = params["input_projection"]
        pre = feats @ p["w"] + p["b"]
        a = elu(pre)
        rows["feats"] = feats
    else:
        gf = inputs.reshape(-1, 1)
        a = log_and_sign(gf[..., None], **cfg.preprocess_options).reshape(gf.shape[0], -1) \
            if cfg.preprocess_name == "LogAndSign" else gf
    if len(cfg.layers) == 0:
        lin = params["linear"]
        d = a @ lin["w"] + lin["b"]
        dd = dx_next.reshape(-1, 1) * dt(cfg.scale)
        if cfg.tanh_output:
            dd = dd * (dt(1) - np.tanh(d) ** 2)
        rows.update(act1=a, dd=dd[:, 0], da=dd @ lin["w"].T)
        return None, rows
    (h1p, c1p), (h2p, c2p) = state_prev
    H = h1p.shape[1]

    def fwd(x, h, c, p):
        z = np.concatenate([x, h], 1) @ p["w_gates"] + p["b_gates"]
        i, j, f, o = sigmoid(z[:, :H]), np.tanh(z[:, H:2 * H]), sigmoid(z[:, 2 * H:3 * H] + dt(1)), sigmoid(z[:, 3 * H:])
        cn = f * c + i * j
        return i, j, f, o, cn, np.tanh(cn)

    i1, j1, f1, o1, c1, tc1 = fwd(a, h1p, c1p, params["lstm_1"])
    h1 = tc1 * o1
    i2, j2, f2, o2, c2, tc2 = fwd(h1, h2p, c2p, params["lstm_2"])
    h2 = tc2 * o2
    lin = params["linear"]
    d = h2 @ lin["w"] + lin["b"]
    dd = dx_next.reshape(-1, 1) * dt(cfg.scale)
    if cfg.tanh_output:
        dd = dd * (dt(1) - np.tanh(d) ** 2)
    dh1_in, dc1_in, dh2_in, dc2_in = carry_in

    def bwd(dh, dc_in, i, j, f, o, tc, c_prev, p, n_in):
        dc = dc_in + dh * o * (dt(1) - tc * tc)
        dz = np.concatenate([dc * j * i * (dt(1) - i), dc * i * (dt(1) - j * j), dc * c_prev * f * (dt(1) - f),
                             dh * tc * o * (dt(1) - o)], 1)
        din = dz @ p["w_gates"].T
        return dz, din[:, :n_in], din[:, n_in:], dc * f

    dh2 = dd @ lin["w"].T + dh2_in
    dz2, dh1_from2, dh2_out, dc2_out = bwd(dh2, dc2_in, i2, j2, f2, o2, tc2, c2p, params["lstm_2"], H)
    dh1 = dh1_from2 + dh1_in
    dz1, da, dh1_out, dc1_out = bwd(dh1, dc1_in, i1, j1, f1, o1, tc1, c1p, params["lstm_1"], a.shape[1])
    rows.update(act1=np.concatenate([a, h1p], 1), dz1=dz1, act2=np.concatenate([h1, h2p], 1), dz2=dz2, h2=h2,
                dd=dd[:, 0])
    if cfg.kind == "rnnprop":
        rows["du"] = da * np.where(pre > 0, dt(1), np.exp(np.minimum(pre, dt(0))))
    rows["da"] = da            # adjoint of the network's (preprocessed) input: second_derivatives (preprocess_bwd)
    return (dh1_out, dc1_out, dh2_out, dc2_out), rows


def preprocess_bwd(cfg, g, da):
    """dL/dg [N] of a DM net from the adjoint ``da`` [N, P] of its preprocessed input (what flows back into the optimizee
    gradient when MetaOptimizer.meta_loss(second_derivatives=True) drops the stop_gradient, DM/meta.py:328-329):
    identity: da itself; LogAndSign (DM/preprocess.py:63-70): the log column has slope sign(g) / (k (|g| + eps)) where
    it is not clamped at -1, the sign column slope e^k where |g e^k| < 1."""
    g = g.reshape(-1)
    dt = g.dtype.type
    if cfg.preprocess_name != "LogAndSign":
        return da[:, 0]
    k = dt(cfg.preprocess_options["k"])
    eps = dt(np.finfo(np.float32).eps)
    mag = np.abs(g) + eps
    d_log = np.where(np.log(mag) / k > dt(-1), np.sign(g) / (k * mag), dt(0))
    ek = dt(np.exp(k))
    d_sign = np.where(np.abs(g * ek) < dt(1), ek, dt(0))
    return da[:, 0] * d_log + da[:, 1] * d_sign


def tf_adam_step(var, g, m, v, t, lr=0.01, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer._apply_dense (TF 1.x), fp32: returns (var', m', v')."""
    f = np.float32
    lr_t = f(lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
    m = f(beta1) * m + f(1.0 - beta1) * g
    v = f(beta2) * v + f(1.0 - beta2) * g * g
    return var - lr_t * m / (np.sqrt(v) + f(eps)), m, v
