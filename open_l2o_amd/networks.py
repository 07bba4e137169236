"""Learning 2 Learn meta-optimizer networks -- the reference's ``DM/networks.py``
(DM = /root/reference/Model_Free_L2O/"L2O-DM and L2O-RNNProp"/) API, session-less.

Same names: ``factory``, ``save``, ``Network``, ``StandardDeepLSTM``,
``CoordinateWiseDeepLSTM``, ``RNNprop``, ``KernelDeepLSTM``, ``Sgd``, ``Adam``.
A network object owns its weights as host ndarrays in the exact ``.l2l`` layout
of ``networks.save`` (DM/networks.py:47-62):
``{module_name: {variable_name: ndarray}}`` with Sonnet's module / variable names
(``lstm_1/w_gates`` ...), so weights trained by the TF reference drop in and
vice-versa.  All arithmetic happens in the HIP kernels; ``net(inputs, state)``
runs ``l2o_cwlstm_step`` eagerly on device tensors.
"""
from __future__ import annotations

import abc
import collections
import sys

import dill as pickle
import numpy as np

from . import _abi, _engine
from ._engine import NetSpec

_rng = np.random.default_rng(0)


def set_random_seed(seed):
    """Seed of the weight initialisers (the reference uses tf.set_random_seed)."""
    global _rng
    _rng = np.random.default_rng(seed)


def factory(net, net_options=(), net_path=None):
    """Network factory.  DM/networks.py:34-44."""
    net_class = getattr(sys.modules[__name__], net)
    net_options = dict(net_options)
    if net_path:
        with open(net_path, "rb") as f:
            net_options["initializer"] = pickle.load(f)
    return net_class(**net_options)


def save(network, sess=None, filename=None):
    """Save the variables contained by a network to disk.  DM/networks.py:47-62."""
    to_save = collections.defaultdict(dict)
    for module_name, variables in network.variables.items():
        for variable_name, value in variables.items():
            to_save[module_name][variable_name] = np.array(value, copy=True)
    if filename:
        with open(filename, "wb") as f:
            pickle.dump(to_save, f)
    return to_save


class Network(abc.ABC):
    """Base class for meta-optimizer networks.  DM/networks.py:65-72."""

    #: {module: {variable: ndarray}} (empty for Sgd / Adam)
    variables = {}

    @abc.abstractmethod
    def initial_state_for_inputs(self, inputs, **kwargs):
        """Initial state given inputs."""


# ---------------------------------------------------------------------------
# initializer plumbing, DM/networks.py:75-151
# ---------------------------------------------------------------------------
def _truncated_normal(shape, fan_in):
    out = _rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = _rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out / np.sqrt(fan_in)).astype(np.float32)


def _materialize(initializer, shape, default):
    """_convert_to_initializer (DM/networks.py:75-95) applied to one variable."""
    if initializer is None:
        return default()
    if isinstance(initializer, str):
        if initializer == "zeros":
            return np.zeros(shape, np.float32)
        if initializer == "ones":
            return np.ones(shape, np.float32)
        raise ValueError("unknown initializer %r" % (initializer,))
    if isinstance(initializer, np.ndarray):
        arr = np.asarray(initializer, np.float32)
        if arr.shape != tuple(shape):
            raise ValueError("initializer shape %r != variable shape %r" % (arr.shape, tuple(shape)))
        return arr.copy()
    if callable(initializer):
        return np.asarray(initializer(shape), np.float32).reshape(shape)
    raise ValueError("unsupported initializer %r" % (type(initializer),))


def _get_layer_initializers(initializers, layer_name, fields):
    """DM/networks.py:126-151 (+ _get_initializers :98-123): per-field initializer dict
    for one layer; fields without an entry fall back to the Sonnet default."""
    if initializers is None:
        return {}
    if isinstance(initializers, dict) and layer_name in initializers:
        initializers = initializers[layer_name]
    result = {}
    for f in fields:
        if isinstance(initializers, dict):
            if f in initializers:
                result[f] = initializers[f]
        else:
            result[f] = initializers
    return result


class StandardDeepLSTM(Network):
    """LSTM layers with a Linear layer on top.  DM/networks.py:154-236.

    Only the coordinate-wise subclasses (output_size == 1) have HIP kernels."""

    _kind = _abi.NET_CW
    _raw_inputs = 1

    def __init__(self, output_size, layers, preprocess_name="identity", preprocess_options=None,
                 scale=1.0, initializer=None, name="deep_lstm", tanh_output=False):
        if output_size != 1:
            raise NotImplementedError("only coordinate-wise (output_size=1) networks are implemented; "
                                      "KernelDeepLSTM / StandardDeepLSTM are outside the hot path")
        self.name = name
        self._output_size = output_size
        self._scale = scale
        self._preprocess_name = preprocess_name
        self.tanh_output = tanh_output
        self._layers = tuple(int(h) for h in layers)
        preprocess_options = dict(preprocess_options or {})
        variables = collections.OrderedDict()

        if preprocess_name == "fc":                                   # :180-183
            dim = int(preprocess_options["dim"])
            init = _get_layer_initializers(initializer, "input_projection", ("w", "b"))
            variables["input_projection"] = {
                "w": _materialize(init.get("w"), (self._raw_inputs, dim),
                                  lambda: _truncated_normal((self._raw_inputs, dim), self._raw_inputs)),
                "b": _materialize(init.get("b"), (dim,), lambda: np.zeros((dim,), np.float32)),
            }
            pre, in_dim, k = _abi.PRE_FC_ELU, dim, 0.0
        elif preprocess_name == "LogAndSign":                         # :184-186
            pre, in_dim, k = _abi.PRE_LOGSIGN, 2 * self._raw_inputs, float(preprocess_options["k"])
        elif preprocess_name == "identity":                           # :187-188 (tf.identity)
            pre, in_dim, k = _abi.PRE_IDENTITY, self._raw_inputs, 0.0
        else:
            raise ValueError("preprocess_name %r is not implemented (identity | LogAndSign | fc)"
                             % (preprocess_name,))
        if self._kind == _abi.NET_RNNPROP and pre != _abi.PRE_FC_ELU:
            raise NotImplementedError("RNNprop is implemented with preprocess_name='fc' (DM/util.py:251-263)")
        if self._kind == _abi.NET_CW and pre == _abi.PRE_FC_ELU:
            raise NotImplementedError("CoordinateWiseDeepLSTM with preprocess_name='fc' is not implemented")

        size_in = in_dim
        for i, size in enumerate(self._layers, start=1):             # :192-197
            lname = "lstm_{}".format(i)
            init = _get_layer_initializers(initializer, lname, ("w_gates", "b_gates"))
            shape_w = (size_in + size, 4 * size)
            variables[lname] = {
                "w_gates": _materialize(init.get("w_gates"), shape_w,
                                        lambda s=shape_w: _truncated_normal(s, s[0])),
                "b_gates": _materialize(init.get("b_gates"), (4 * size,),
                                        lambda n=4 * size: np.zeros((n,), np.float32)),
            }
            size_in = size
        init = _get_layer_initializers(initializer, "linear", ("w", "b"))   # :202-203
        variables["linear"] = {
            "w": _materialize(init.get("w"), (size_in, output_size),
                              lambda: _truncated_normal((size_in, output_size), size_in)),
            "b": _materialize(init.get("b"), (output_size,), lambda: np.zeros((output_size,), np.float32)),
        }
        self._host_stale = False
        self.variables = variables
        self.spec = NetSpec(kind=self._kind, preprocess=pre, layers=self._layers, scale=float(scale),
                            tanh_output=bool(tanh_output), logsign_k=k)
        self._wpack = None
        self._wpack_engine = None

    # -- host view of the weights -------------------------------------------
    # The .l2l dict {module: {variable: ndarray}} is the master copy until the meta-step runs on the
    # device (MetaOptimizer.meta_minimize with the HIP engine: Adam + re-pack without a host round
    # trip); from then on the flat device buffer of device_weights() is, and the dict is refreshed from
    # it the next time somebody reads it (save, assign, tests).
    @property
    def variables(self):
        if self._host_stale:
            self._pull_from_device()
        return self._variables

    @variables.setter
    def variables(self, value):
        self._variables = value
        self._host_stale = False
        self._wversion = getattr(self, "_wversion", 0) + 1

    def _pull_from_device(self):
        flat = self._wdev_engine.to_numpy(self._wdev_buf)
        for k, (o, shp) in self._wdev_offs.items():
            mod, var = self._wdev_names[k]
            self._variables[mod][var] = flat[o:o + int(np.prod(shp))].reshape(shp).copy()
        self._host_stale = False

    def mark_device_updated(self):
        """The flat device buffer (and the packed copy) were updated in place: the host dict is stale."""
        self._host_stale = True
        self._wversion = getattr(self, "_wversion", 0) + 1     # (every cache derived from the weights keys on this)

    # -- device weights ----------------------------------------------------
    def wpack(self, engine):
        """Device copy of the weights in MFMA-fragment order (cached until invalidated); for a generic `layers`
        tuple the Sonnet-layout device weights behind a struct l2o_gen_net."""
        if self.spec.generic and hasattr(engine, "gen_net"):
            if self._wpack is None or self._wpack_engine is not engine:
                self._wpack, self._wpack_engine = engine.gen_net(self.spec, self.variables), engine
            return self._wpack
        if self._wpack is None or self._wpack_engine is not engine:
            if hasattr(engine, "upload"):                 # persistent device buffer, pinned staging
                self._wpack = engine.pack_weights(self.spec, self.variables, key=(id(self), "wpack"))
            else:
                self._wpack = engine.pack_weights(self.spec, self.variables)
            self._wpack_engine = engine
        return self._wpack

    def assign(self, module_name, variable_name, value):
        """Overwrite one weight (used by MetaOptimizer.restore and the Adam meta-step)."""
        cur = self.variables[module_name][variable_name]          # (refreshes a stale host copy first)
        self._variables[module_name][variable_name] = np.asarray(value, np.float32).reshape(cur.shape).copy()
        self._wpack = None
        self._wdev = None
        self._wversion = getattr(self, "_wversion", 0) + 1

    def device_weights(self, engine):
        """Device copies of the weights in their Sonnet layouts, keyed like struct
        l2o_net_weights (the BPTT kernel reads them unpacked)."""
        if getattr(self, "_wdev", None) is None or self._wdev_engine is not engine:
            v = self.variables
            names = {"w_gates1": ("lstm_1", "w_gates"), "b_gates1": ("lstm_1", "b_gates"),
                     "w_gates2": ("lstm_2", "w_gates"), "b_gates2": ("lstm_2", "b_gates"),
                     "w_lin": ("linear", "w"), "b_lin": ("linear", "b"),
                     "w_fc": ("input_projection", "w"), "b_fc": ("input_projection", "b")}
            # one host-to-device copy: the tensors are 16-byte aligned views of a single buffer
            parts, offs, off = [], {}, 0
            for k, (m, n) in names.items():
                if m in v:
                    a = np.ascontiguousarray(v[m][n], np.float32)
                    offs[k] = (off, a.shape)
                    parts.append(a.reshape(-1))
                    pad = (-a.size) % 4
                    if pad:
                        parts.append(np.zeros(pad, np.float32))
                    off += a.size + pad
            flat = np.concatenate(parts)
            buf = engine.upload((id(self), "wdev"), flat) if hasattr(engine, "upload") else engine.tensor(flat)
            self._wdev = {k: buf[o:o + int(np.prod(shp))].view(*shp) for k, (o, shp) in offs.items()}
            self._wdev_buf, self._wdev_offs, self._wdev_names = buf, offs, names
            if len(self.spec.layers):
                self._wdev["wpack"] = self.wpack(engine)      # selects the matrix-core BPTT kernel
            self._wdev_engine = engine
        return self._wdev

    # -- eager call: net(inputs, prev_state) -> (delta, next_state) ----------
    def _panel(self, inputs):
        n = int(np.prod(inputs.shape)) if inputs.dim() > 0 else 1
        return n

    def initial_state_for_inputs(self, inputs, **kwargs):            # :234-236
        from .meta import PackedState
        engine = kwargs.pop("engine", None) or _engine.default_engine()
        return PackedState.zeros(engine, 1, self._panel(inputs), self._layers)

    def __call__(self, inputs, prev_state):
        """Functional form of ``_build`` (DM/networks.py:207-232): returns
        ``(delta shaped like inputs, next_state)`` without modifying its arguments."""
        engine = prev_state.engine
        n = self._panel(inputs)
        g = inputs.reshape(1, n).contiguous()
        delta = engine.zeros(1, n)
        nxt = prev_state.clone()
        engine.lstm_step(self.spec, self.wpack(engine), g, None, None, 0.0, 0.0, nxt.packed, delta, 1, n)
        return delta.reshape(inputs.shape), nxt


class CoordinateWiseDeepLSTM(StandardDeepLSTM):
    """Coordinate-wise `DeepLSTM`.  DM/networks.py:239-276."""

    def __init__(self, name="cw_deep_lstm", **kwargs):
        super(CoordinateWiseDeepLSTM, self).__init__(1, name=name, **kwargs)


class RNNprop(StandardDeepLSTM):
    """DM/networks.py:279-300: inputs are the pair (m~, g~)."""

    _kind = _abi.NET_RNNPROP
    _raw_inputs = 2

    def __init__(self, name="RNNprop", **kwargs):
        super(RNNprop, self).__init__(1, name=name, **kwargs)

    def __call__(self, m, g, prev_state):
        """The plugin contract of DM/networks.py:287-295: ``net(m, g, prev_state) -> (delta shaped like g,
        next_state)`` on the pre-normalised pair (m~, g~) -- no moments are read or written here (the unroll
        kernels consume the raw gradient and carry them; this is the eager form).  One launch of
        l2o_cwlstm_step_generic with direct inputs; the arguments are not modified."""
        from .meta import PackedState
        import torch
        engine = prev_state.engine
        n = self._panel(g)
        gen = self.__dict__.get("_gen_direct")
        # keyed on the weights VERSION: assign() / the device meta-step mutate the dict in place (its identity never
        # changes), and an eager call after restore() or a training step must see the new weights
        if gen is None or gen[0] is not engine or gen[2] != self._wversion:
            gen = self._gen_direct = (engine, engine.gen_net(self.spec, self.variables, direct=True), self._wversion)
        if prev_state.generic:
            st = prev_state.packed.clone()
        else:                                              # (20, 20): tile-major packed -> per-layer [N, H] -> back
            st = torch.cat([t.reshape(-1) for t in engine.state_unpack(prev_state.packed, prev_state.B, prev_state.D)])
        delta = engine.zeros(n)
        engine.lstm_step_generic(self.spec, gen[1], g.reshape(n).contiguous(), m.reshape(n).contiguous(), None, None,
                                 0.0, 0.0, st, delta, n)
        if prev_state.generic:
            nxt = PackedState(engine, st, prev_state.B, prev_state.D, self._layers)
        else:
            H = 20
            parts = [st[k * n * H:(k + 1) * n * H].view(n, H) for k in range(4)]
            nxt = PackedState(engine, engine.state_pack(*parts, prev_state.B, prev_state.D), prev_state.B, prev_state.D,
                              self._layers)
        return delta.reshape(g.shape), nxt


class KernelDeepLSTM(StandardDeepLSTM):
    """`DeepLSTM` for convolutional filters (DM/networks.py:303-351): out of scope."""

    def __init__(self, kernel_shape, name="kernel_deep_lstm", **kwargs):
        raise NotImplementedError("KernelDeepLSTM (conv-filter optimizee nets) is outside the hot path")


class Sgd(Network):
    """Identity network which acts like SGD.  DM/networks.py:354-371."""

    def __init__(self, learning_rate=0.001, name="sgd"):
        self.name = name
        self._learning_rate = learning_rate
        self.variables = {}

    def __call__(self, inputs, _):
        return -self._learning_rate * inputs, []

    def initial_state_for_inputs(self, inputs, **kwargs):
        return []


class Adam(Network):
    """Adam algorithm (https://arxiv.org/pdf/1412.6980v8.pdf).  DM/networks.py:382-420.

    A hand-designed baseline net, not part of the learned-optimizer hot path: plain
    elementwise tensor expressions on the device tensors."""

    def __init__(self, learning_rate=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8, name="adam"):
        self.name = name
        self._learning_rate = learning_rate
        self._beta1 = beta1
        self._beta2 = beta2
        self._epsilon = epsilon
        self.variables = {}

    def __call__(self, g, prev_state):
        b1, b2 = self._beta1, self._beta2
        g_shape = g.shape
        g = g.reshape(-1, 1)
        t, m, v = prev_state
        t_next = t + 1
        m_next = b1 * m + (1 - b1) * g
        m_hat = m_next / (1 - b1 ** t_next)
        v_next = b2 * v + (1 - b2) * g * g
        v_hat = v_next / (1 - b2 ** t_next)
        update = -self._learning_rate * m_hat / (v_hat.sqrt() + self._epsilon)
        return update.reshape(g_shape), (t_next, m_next, v_next)

    def initial_state_for_inputs(self, inputs, **kwargs):
        n = int(np.prod(inputs.shape)) if inputs.dim() > 0 else 1
        return (0.0, inputs.new_zeros((n, 1)), inputs.new_zeros((n, 1)))
