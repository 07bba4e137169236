"""Learning to learn (meta) optimizer -- the reference's ``DM/meta.py`` API
(DM = /root/reference/Model_Free_L2O/"L2O-DM and L2O-RNNProp"/), session-less.

``MetaOptimizer(**net_config).meta_loss(make_loss, len_unroll, net_assignments)``
returns ``MetaLoss(loss, update, reset, fx, x)`` exactly like DM/meta.py:269-396;
the members are *fetch handles* that ``session.Session.run`` evaluates, so the
reference's harness (``util.run_epoch`` / ``run_eval_epoch``,
``evaluate_dm.main``) runs unchanged in structure:

    sess.run(reset)                    re-initialise x, problem data, LSTM state
    sess.run([fx, update])             one unroll of ``len_unroll`` steps from the
                                       current variables; ``update`` carries x_T and
                                       the LSTM state (and RNNProp m, v) over

One ``sess.run`` = ONE launch of the fused persistent HIP kernel ``l2o_unroll``
when the (problem, net) pair has one (Quadratic / Lasso / Rastrigin with up to
128 parameters per problem and a (20, 20) coordinate-wise LSTM), otherwise
``len_unroll`` x {``l2o_problem_fg``, ``l2o_cwlstm_step``} launches.  Either way
every flop of the path runs in HIP kernels behind the C ABI of
``include/l2o_abi.h``; this module only owns buffers and control flow.

Multi-GPU: with ``torch.distributed`` initialised the problem batch is sharded
by contiguous slices over ranks, the 1/B of the loss mean stays the GLOBAL batch
(DM/problems.py:99, 131, 211) and the only collective is one all-reduce (RCCL)
of the ``len_unroll + 1`` partial losses per unroll.
"""
from __future__ import annotations

import collections
import os

import numpy as np
import torch

from . import _abi, _engine, networks
from ._engine import ProblemDesc

MetaLoss = collections.namedtuple("MetaLoss", "loss, update, reset, fx, x")     # DM/meta.py:158
MetaStep = collections.namedtuple("MetaStep", "step, update, reset, fx, x")     # DM/meta.py:159

_rng = np.random.default_rng(0)


def set_random_seed(seed):
    """Seed for the optimizee initialisers (x0, W, y ...) and the network weights --
    the analogue of ``tf.set_random_seed`` (DM/evaluate_dm.py:51-52)."""
    global _rng
    _rng = np.random.default_rng(seed)
    networks.set_random_seed(None if seed is None else seed + 1)


# ---------------------------------------------------------------------------
# handles
# ---------------------------------------------------------------------------
class Fetch(object):
    """Something ``Session.run`` can evaluate: (graph, key)."""

    def __init__(self, graph, key, name=None):
        self.graph, self.key, self.name = graph, key, name or key

    def __repr__(self):
        return "<Fetch %s>" % self.name


class Placeholder(object):
    """``tf.placeholder`` / ``placeholder_with_default`` analogue (scale, step)."""

    def __init__(self, name, shape, default=None, dtype="float32"):
        self.name, self.shape, self.default, self.dtype = name, tuple(shape), default, dtype

    def __repr__(self):
        return "<Placeholder %s %s>" % (self.name, self.shape)


class Variable(object):
    """An optimizee variable (``tf.Variable`` analogue): ``name`` ("x:0"), ``shape`` (the
    GLOBAL shape), ``value`` (device tensor holding this rank's batch shard)."""

    def __init__(self, decl, graph, sharded):
        self.decl = decl
        self.name = decl.name + ":0"
        self.shape = decl.shape
        self.trainable = decl.trainable
        self._graph = graph
        self.sharded = sharded
        self.value = None

    def _local(self, arr):
        arr = np.asarray(arr, np.float32).reshape(self.shape)
        if self.sharded:
            lo, hi = self._graph.shard
            arr = arr[lo:hi]
        return np.ascontiguousarray(arr)

    def initial_value(self):
        init = self.decl.initializer
        shape = self.shape
        if init is None or init[0] == "zeros":
            arr = np.zeros(shape, np.float32)
        elif init[0] == "ones":
            arr = np.ones(shape, np.float32)
        elif init[0] == "normal":
            arr = _rng.standard_normal(shape, dtype=np.float32) * np.float32(init[2]) + np.float32(init[1])
        elif init[0] == "uniform":
            arr = _rng.random(shape, dtype=np.float32) * np.float32(init[2] - init[1]) + np.float32(init[1])
        elif init[0] == "constant":
            arr = np.broadcast_to(init[1], shape).astype(np.float32)
        else:
            raise ValueError("unknown initializer %r" % (init,))
        return arr

    def initialize(self):
        """(Re)sample the variable (MetaLoss.reset, DM/meta.py:379-383 runs the tf initializers -- device ops there too).
        Random initializers are drawn ON THE DEVICE when the engine can (HipEngine.sample: a torch generator seeded from
        the stream of set_random_seed): the host draw + upload of config 2's 128 x 128 x 128 matrix batch was 4 ms per
        reset, twice the five 20-step training unrolls of an epoch.  Every rank draws the GLOBAL array from the same
        seed and keeps its shard (the ranks together hold the problem batch a single process would).
        L2O_HOST_SAMPLING=1: the NumPy draw."""
        eng = self._graph.engine
        init = self.decl.initializer
        if (init is not None and init[0] in ("normal", "uniform") and hasattr(eng, "sample")
                and not os.environ.get("L2O_HOST_SAMPLING")):
            seed = int(_rng.integers(0, 2 ** 62))
            t = eng.sample(init[0], tuple(self.shape), float(init[1]), float(init[2]), seed)
            if self.sharded:
                lo, hi = self._graph.shard
                t = t[lo:hi].clone()       # (a slice view would keep the whole global draw alive)
            self.value = t
            return
        self.value = eng.tensor(self._local(self.initial_value()))

    def load(self, value, session=None):
        """tf.Variable.load: assign a value -- the GLOBAL shape, or (sharded) this rank's shard, i.e.
        what ``eval`` / ``sess.run(var)`` returned."""
        value = np.asarray(value, np.float32)
        if self.sharded and value.shape == self._graph._local_shape(self) and value.shape != tuple(self.shape):
            self.value = self._graph.engine.tensor(np.ascontiguousarray(value))
        else:
            self.value = self._graph.engine.tensor(self._local(value))
        self._graph.__dict__.pop("_fast_unrolls", None)      # prepared calls point into the old buffer

    def eval(self, session=None):
        """This rank's shard as an ndarray."""
        return self._graph.engine.to_numpy(self.value)

    def __repr__(self):
        return "<Variable %s %s>" % (self.name, self.shape)


class PackedState(object):
    """LSTM state of one variable in the packed tile-major device layout
    (``l2o_state_floats``); ``unpack()`` gives the reference structure
    ``((hidden_1, cell_1), (hidden_2, cell_2))`` with [N, H] arrays
    (DM/networks.py:234-236; index [l][0] = hidden, [l][1] = cell)."""

    def __init__(self, engine, packed, B, D, layers):
        self.engine, self.packed, self.B, self.D, self.layers = engine, packed, B, D, tuple(int(h) for h in layers)

    @property
    def generic(self):
        """layers other than (20, 20): the per-layer [N, H] layout of l2o_cwlstm_step_generic, not the tile-major one."""
        return len(self.layers) > 0 and self.layers != (20, 20)

    @classmethod
    def zeros(cls, engine, B, D, layers):
        layers = tuple(int(h) for h in layers)
        if len(layers) == 0:
            return cls(engine, None, B, D, layers)
        if layers != (20, 20):
            return cls(engine, engine.zeros(2 * B * D * sum(layers)), B, D, layers)
        return cls(engine, engine.state_alloc(B, D), B, D, layers)

    def clone(self):
        return PackedState(self.engine, None if self.packed is None else self.packed.clone(), self.B, self.D,
                           self.layers)

    def zero_(self):
        if self.packed is not None:
            self.packed.zero_()

    def unpack(self):
        if self.packed is None:
            return ()
        if self.generic:
            N, out, off = self.B * self.D, [], 0
            for H in self.layers:
                out.append((self.packed[off:off + N * H].view(N, H), self.packed[off + N * H:off + 2 * N * H].view(N, H)))
                off += 2 * N * H
            return tuple(out)
        h1, c1, h2, c2 = self.engine.state_unpack(self.packed, self.B, self.D)
        return ((h1, c1), (h2, c2))

    def load(self, state):
        t = self.engine.tensor
        if self.generic:
            self.packed = torch.cat([t(a).reshape(-1) for hc in state for a in hc])
            return
        (h1, c1), (h2, c2) = state
        self.packed = self.engine.state_pack(t(h1), t(c1), t(h2), t(c2), self.B, self.D)


# ---------------------------------------------------------------------------
# net construction, DM/meta.py:162-216
# ---------------------------------------------------------------------------
def _make_nets(variables, config, net_assignments):
    """Creates the optimizer networks; returns (nets, keys, subsets).  DM/meta.py:162-216."""
    name_to_index = dict((v.name.split(":")[0], i) for i, v in enumerate(variables))
    if net_assignments is None:
        if len(config) != 1:
            raise ValueError("Default net_assignments can only be used if there is "
                             "a single net config.")
        key = next(iter(config))
        kwargs = config[key]
        net = networks.factory(**kwargs)
        nets = {key: net}
        keys = [key]
        subsets = [list(range(len(variables)))]
    else:
        nets = {}
        keys = []
        subsets = []
        for key, names in net_assignments:
            if key in nets:
                raise ValueError("Repeated netid in net_assigments.")
            nets[key] = networks.factory(**config[key])
            subset = [name_to_index[name] for name in names]
            keys.append(key)
            subsets.append(subset)
    return nets, keys, subsets


_DEFAULT_CONFIG = {
    "coordinatewise": {
        "net": "CoordinateWiseDeepLSTM",
        "net_options": {
            "layers": (20, 20),
            "preprocess_name": "LogAndSign",
            "preprocess_options": {"k": 5},
            "scale": 0.01,
        }}}


class _DevGrad(object):
    """A weight gradient that stays on the device (the meta-step consumes it there); NumPy sees it as an
    array (copied to the host on demand: tests, the host Adam path)."""
    __slots__ = ("t", "src")

    def __init__(self, t, src=None):
        self.t = t
        self.src = src          # (G, row0, col0): t is the block G[row0:row0 + r, col0:col0 + c] of a contraction result

    def __array__(self, dtype=None, copy=None):
        a = self.t.detach().cpu().numpy()
        return a if dtype is None else a.astype(dtype)


class _LazyHost(object):
    """res["x"]: the final iterates, copied device -> host when (and only when) one is fetched
    (every .cpu() is a stream synchronisation; sess.run([fx, update, step]) does not ask for x)."""

    def __init__(self, engine, tensors, shapes):
        self._e, self._t, self._s = engine, list(tensors), list(shapes)

    def __len__(self):
        return len(self._t)

    def __getitem__(self, j):
        return self._e.to_numpy(self._t[j]).reshape(self._s[j])

    def __iter__(self):
        return (self[j] for j in range(len(self._t)))


def _term_vars(term):
    """The trainable variable declarations a loss term is a function of (one for the analytic
    problems, four for problems.mnist)."""
    return (term.var,) if hasattr(term.var, "initializer") else tuple(term.var)


def _world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def synced_scale(shape, bound):
    """exp(U[-bound, bound]) of the given (GLOBAL) shape from ``np.random`` (DM/util.py:44), identical on
    every rank: drawn on rank 0 and broadcast when torch.distributed is initialised."""
    arr = np.exp(np.random.uniform(-bound, bound, size=tuple(shape)))
    if _world()[1] > 1:
        import torch.distributed as dist
        box = [arr]
        dist.broadcast_object_list(box, src=0)
        arr = box[0]
    return arr


def local_slice(var, arr):
    """This rank's part of a global-shape array that belongs to ``var`` (the whole array when unsharded)."""
    return var._local(arr).reshape(var._graph._local_shape(var)) if isinstance(var, Variable) else np.asarray(arr)


# ---------------------------------------------------------------------------
# the unroll "graph"
# ---------------------------------------------------------------------------
class _Slot(object):
    """One (net, variable) pairing: the LSTM state (and RNNProp moments) of a variable."""

    def __init__(self, key, net, var_index):
        self.key, self.net, self.var_index = key, net, var_index
        self.state = None      # PackedState | Adam tuple | []
        self.m = None
        self.v = None


class UnrollGraph(object):
    """Everything one ``meta_loss`` call builds (DM/meta.py:293-394)."""

    def __init__(self, optimizer, make_loss, len_unroll, net_assignments, rnnprop=False, beta1=0.95,
                 beta2=0.95, engine=None):
        self.engine = engine or _engine.default_engine()
        self.len_unroll = int(len_unroll)
        self.rnnprop = rnnprop
        self.beta1, self.beta2 = float(beta1), float(beta2)
        self.rank, self.world = _world()

        loss = make_loss()                                            # _get_variables, :293
        self.terms = list(loss.terms)
        decls = list(loss.variables)
        batched_kinds = (_abi.PROB_QUADRATIC, _abi.PROB_LASSO, _abi.PROB_RASTRIGIN, _abi.PROB_SQUARE_COS)
        self.sharded = self.world > 1 and all(t.kind in batched_kinds for t in self.terms)
        B_global = decls[0].shape[0] if self.sharded else None
        if self.sharded:
            if any(d.shape[0] != B_global for d in decls if not getattr(d, "shared", False)):
                raise ValueError("cannot shard: variables disagree on the batch dimension")
            if B_global % self.world:
                raise ValueError("batch_size %d is not divisible by world size %d" % (B_global, self.world))
            per = B_global // self.world
            self.shard = (self.rank * per, (self.rank + 1) * per)
        else:
            self.shard = None
        by_name = {}
        self.x, self.constants = [], []
        for d in decls:
            v = Variable(d, self, self.sharded and not getattr(d, "shared", False))
            by_name[d.name] = v
            (self.x if d.trainable else self.constants).append(v)
        self._by_name = by_name

        self.nets, self.net_keys, self.subsets = _make_nets(self.x, optimizer._config, net_assignments)
        optimizer._nets = self.nets
        for net in self.nets.values():
            if isinstance(net, networks.StandardDeepLSTM):
                net.spec.beta1, net.spec.beta2 = self.beta1, self.beta2
                if rnnprop != (net.spec.kind == _abi.NET_RNNPROP):
                    raise ValueError("RNNprop networks need the RNNProp MetaOptimizer (meta_rnnprop_eval / "
                                     "meta_rnnprop_train) and vice versa")

        # placeholders of the train / RNNProp forks (DM/meta_dm_train.py:336-338,
        # DM/meta_rnnprop_eval.py: scale + step)
        self.scale = [Placeholder(v.name[:-2] + "_scale", v.shape, default=None) for v in self.x]
        self.step = Placeholder("step", (), default=None, dtype="int32")

        # term of each trainable variable
        self.term_of = {}
        for t in self.terms:
            for tv in _term_vars(t):
                if tv.name in self.term_of:
                    raise ValueError("variable %s appears in two loss terms" % tv.name)
                self.term_of[tv.name] = t
        for v in self.x:
            if v.decl.name not in self.term_of:
                raise ValueError("no loss term for variable %s" % v.name)

        self.slots = []
        for subset, key in zip(self.subsets, self.net_keys):
            for j in subset:
                self.slots.append(_Slot(key, self.nets[key], j))
        self._initialized = False
        self._fx_cache = {}
        self.last_path = None
        self.second_derivatives = False

    # -- geometry helpers ----------------------------------------------------
    def _panel_shape(self, var):
        """[B_local, D] view of a variable for the kernels."""
        term = self.term_of[var.decl.name]
        if term.kind in (_abi.PROB_SIMPLE, _abi.PROB_MLP):
            return 1, int(np.prod(var.shape)) if len(var.shape) else 1
        B = var.shape[0]
        if self.sharded:
            B = self.shard[1] - self.shard[0]
        return B, int(np.prod(var.shape[1:]))

    def _desc(self, var, x_scale):
        term = self.term_of[var.decl.name]
        B, D = self._panel_shape(var)
        if term.kind == _abi.PROB_SIMPLE:
            return ProblemDesc(_abi.PROB_SIMPLE, 1, 1, D, x_scale=x_scale)
        Bg = var.shape[0]
        W = self._by_name[term.consts["W"].name].value
        y = self._by_name[term.consts["y"].name].value
        C = self._by_name[term.consts["C"].name].value if "C" in term.consts else None
        if term.kind == _abi.PROB_SQUARE_COS:
            # sum_i (wcos c)_i = sum_j colsum_j(wcos) c_j: the kernels take the column sums
            # (a per-reset constant of the problem instance) in the C slot
            wc = self._by_name[term.consts["wcos"].name].value
            cache = self.__dict__.setdefault("_colsum", {})
            if cache.get("src") is not wc:
                cache["src"], cache["val"] = wc, wc.sum(dim=1).contiguous()
            C = cache["val"]
        M = term.consts["W"].shape[1]
        return ProblemDesc(term.kind, B, Bg, D, M=M, l1=term.hyper.get("l1", 0.0),
                           alpha=term.hyper.get("alpha", 0.0), W=W, y=y, C=C, x_scale=x_scale,
                           w_shared=bool(getattr(term.consts["W"], "shared", False)))

    # -- reset / init (DM/meta.py:379-383; RNNProp :559-566) -------------------
    def reset(self):
        eng = self.engine
        for v in self.x + self.constants:
            v.initialize()
        for s in self.slots:
            var = self.x[s.var_index]
            B, D = self._panel_shape(var)
            if isinstance(s.net, networks.StandardDeepLSTM):
                s.state = PackedState.zeros(eng, B, D, s.net.spec.layers)
                if self.rnnprop:
                    s.m, s.v = eng.zeros(B, D), eng.zeros(B, D)
            else:
                s.state = s.net.initial_state_for_inputs(var.value)
        self._initialized = True
        self.__dict__.pop("_fast_unrolls", None)         # prepared calls hold the old x / state buffers
        self.__dict__.pop("_hip_graphs", None)           # captured launch sequences hold the old pointers
        self.__dict__.pop("_mlp_idx", None)

    def _ensure_init(self):
        if not self._initialized:
            self.reset()

    # -- execution -------------------------------------------------------------
    def _fused_ok(self, descs, record=False):
        if len(self.x) != 1 or len(self.slots) != 1 or len(self.terms) != 1 or descs[0] is None:
            return False
        s = self.slots[0]
        if not isinstance(s.net, networks.StandardDeepLSTM) or self.terms[0].weight != 1.0:
            return False
        if os.environ.get("L2O_DISABLE_FUSED"):
            return False
        return self.engine.unroll_supported(s.net.spec, descs[0], record=record)

    def execute(self, feed, commit):
        """Run one unroll from the current variables.  Returns dict(loss, fx, x) on the host."""
        fx, xs = self.launch(feed, commit)
        eng = self.engine
        T = self.len_unroll
        self.wait_fx()
        fused = self.last_path in ("fused", "mlp_unroll") and hasattr(eng, "check_unroll_status")
        if fused and hasattr(eng, "prefetch_unroll_status"):
            eng.prefetch_unroll_status()                 # (rides on the sync below)
        fx_host = eng.to_numpy(fx)                       # host sync
        if fused:
            self._check_unroll_status()
        x_out = _LazyHost(eng, xs, [self._local_shape(var) for var in self.x])   # copied to the host only if fetched
        return {"loss": np.float32(fx_host.sum(dtype=np.float32)), "fx": np.float32(fx_host[T]),
                "x": x_out, "fx_array": fx_host}

    def deterministic(self):
        """True when an unroll draws nothing at random (no minibatch sampling): then n committed unrolls
        of L steps are exactly one unroll of n * L steps."""
        return all(t.kind != _abi.PROB_MLP for t in self.terms)

    def execute_many(self, n):
        """n consecutive committed unrolls (what util.run_eval_epoch asks for with n sess.run calls,
        DM/util.py:78-89: the evaluation drivers use len_unroll = 1) as ONE unroll of n * len_unroll
        steps -- one launch of the fused kernel where it applies, no per-step host round trip.
        x_{t+1} = x_t + delta_t and the LSTM state carry across the reference's unroll boundaries
        unchanged (MetaLoss.update, DM/meta.py:379-383), and RNNProp's fed `step` = i * L + 1 is the
        running step count (DM/util.py:84-87), so the k-th unroll's fx is entry (k + 1) * L of the long
        unroll's loss array.  Returns those n values (host)."""
        assert n >= 1
        if not self.deterministic():
            return self._execute_many_sampled(n)
        L = self.len_unroll
        self.len_unroll = n * L
        try:
            fx, _ = self.launch({self.step: 1} if self.rnnprop else {}, commit=True)
        finally:
            self.len_unroll = L
        eng = self.engine
        self.wait_fx()
        fx_host = eng.to_numpy(fx)
        if self.last_path == "fused" and hasattr(eng, "check_unroll_status"):
            self._check_unroll_status()
        return [np.float32(fx_host[(k + 1) * L]) for k in range(n)]

    def many_ok(self):
        """execute_many applies: deterministic optimizee, or ONE MLP term stepped by LSTM nets on an engine
        with prepared calls (the sampled form below)."""
        if self.deterministic():
            return True
        self._ensure_init()
        eng = self.engine
        if not hasattr(eng, "prepared_mlp_fg") or os.environ.get("L2O_NO_STEP_PLAN") or self.sharded:
            return False
        states = [s.state for s in self.slots]
        return self._plan_ok(self.slots, states, len(self.x))

    def _execute_many_sampled(self, n):
        """n committed unrolls of the minibatch-sampled MLP optimizee (evaluate_*.py --problem mnist) without a
        host round trip per unroll.  Exactly the loop's computation and the loop's random draws: unroll k
        draws L + 1 minibatches (DM/problems.py:282-286: one per evaluation of the loss), steps on the
        first L and REPORTS the loss of the last one at x_L; the next unroll evaluates x_L again on a new
        draw.  All n (L + 1) index rows are drawn up front in the same order, uploaded once, and the
        3-launch steps go out through prepared calls; one device-to-host copy at the end."""
        self._ensure_init()
        eng = self.engine
        L = self.len_unroll
        term = self.terms[0]
        d = self._mlp_desc(term)
        sampler = term.hyper.get("sampler")
        if sampler is None and hasattr(eng, "sample_int") and not os.environ.get("L2O_HOST_SAMPLING"):
            idx = eng.empty_int(n, L + 1, d.batch)           # drawn on the device, like _draw_minibatches (one call)
            eng.sample_int(idx, d.images.shape[0], int(_rng.integers(0, 2 ** 62)))
        else:
            rows = []
            for _ in range(n):                               # the same draws, in the same order, as n x _draw_minibatches(L)
                if sampler is None:
                    rows.append(_rng.integers(0, d.images.shape[0], size=(L + 1, d.batch)))
                else:
                    rows.append(np.asarray(sampler(L + 1, d.batch, d.images.shape[0])).reshape(L + 1, d.batch))
            idx = eng.int_tensor(np.stack(rows))             # [n, L + 1, batch]
        index_of = {v.decl.name: j for j, v in enumerate(self.x)}
        js = [index_of[tv.name] for tv in _term_vars(term)]
        panels = []
        for v in self.x:
            B, D = self._panel_shape(v)
            panels.append(v.value.view(B, D))
        slots = self.slots
        key = (tuple(p.data_ptr() for p in panels), tuple(s.state.packed.data_ptr() for s in slots),
               tuple(0 if s.m is None else s.m.data_ptr() for s in slots))
        plan = self.__dict__.get("_eval_plan")
        if plan is None or plan["key"] != key:
            grads = [eng.empty(*panels[j].shape) for j in range(len(self.x))]
            groups = {}
            for s in slots:
                j = s.var_index
                B, D = panels[j].shape
                groups.setdefault(id(s.net), (s.net, []))[1].append((grads[j], s.m, s.v, s.state.packed, panels[j], B, D))
            plan = self.__dict__["_eval_plan"] = dict(
                key=key, grads=grads,
                fg=eng.prepared_mlp_fg(d, idx[0, 0], *[panels[j] for j in js], [grads[j] for j in js]),
                f=eng.prepared_mlp_fg(d, idx[0, 0], *[panels[j] for j in js], None),
                lstm=[(net, eng.prepared_lstm_step_multi(net.spec, segs)) for net, segs in groups.values()])
        import ctypes
        wp = {id(net): (net.wpack(eng), ctypes.c_void_p(net.wpack(eng).data_ptr())) for net, _ in plan["lstm"]}
        b1, b2 = float(np.float32(self.beta1)), float(np.float32(self.beta2))
        fxbuf = eng.empty(n * (L + 1))
        fxp, ip, row_bytes = fxbuf.data_ptr(), idx.data_ptr(), 4 * d.batch
        fg, f, lstm = plan["fg"], plan["f"], plan["lstm"]
        e = 0                                                 # evaluation counter = row of idx = slot of fxbuf
        for k in range(n):
            for t in range(L):
                fg(fxp + 4 * e, ip + row_bytes * e)
                step = k * L + t + 1                          # evaluate_rnnprop feeds step = k * L + 1 (DM/util.py:84-87)
                p1, p2 = b1 ** step, b2 ** step
                for net, call in lstm:
                    call(wp[id(net)][1], p1, p2)
                e += 1
            f(fxp + 4 * e, ip + row_bytes * e)
            e += 1
        self.last_path = "steps"
        out = eng.to_numpy(fxbuf).reshape(n, L + 1)
        return [np.float32(out[k, L]) for k in range(n)]

    def launch(self, feed=None, commit=True, events=None, use_graph=False, record=None, restart=None):
        """Enqueue one unroll on the current stream WITHOUT synchronising the host; returns
        (device tensor fx[0..T] -- already all-reduced when sharded --, list of device x_T).
        ``events`` = (start, end) torch.cuda.Event pair recorded around the unroll kernels.
        ``use_graph`` (with commit=True, no fed x-scale): the launch sequence of the step-granular
        path is captured once into a HIP graph per ``step0`` and replayed afterwards, which
        removes the per-launch host cost of its 2..6 x T small kernels."""
        self._ensure_init()
        eng = self.engine
        T = self.len_unroll
        feed = feed or {}
        # ---- fast path: the SAME fused launch as before (same buffers / problem tensors / options) replays a
        # prepared call -- one ctypes call instead of ~0.2 ms of argument building, which is a whole config-2 unroll.
        # Both the committed launch of Session.run([fx, update]) (the product path: evaluate_*.py, util.run_eval_epoch)
        # and the restart= form (an evaluation loop over a ring of problem instances) take it; RNNProp's fed `step`
        # is a call-time argument.  The key is the identity of every object the call points into, and the cache entry
        # keeps ALL of them alive, so an id cannot be recycled behind the key.
        fast_key = None
        if (commit and record is None and events is None and len(self.x) == 1 and len(self.slots) == 1
                and hasattr(eng, "prepared_unroll") and all(ph not in feed for ph in self.scale)
                and isinstance(self.slots[0].state, PackedState) and self.slots[0].state.packed is not None
                and (not self.rnnprop or self.step in feed)):
            s0 = self.slots[0]
            fast_objs = (None if restart is None else restart[0], self.x[0].value, s0.state.packed, s0.m, s0.v,
                         getattr(s0.net, "_wpack", None)) + tuple(v.value for v in self.constants)
            fast_key = (T, restart is not None, _abi.options_word()) + tuple(id(o) for o in fast_objs)
            ent = self.__dict__.setdefault("_fast_unrolls", {}).get(fast_key)
            if ent is not None:
                ring = self._fx_cache[T]
                i = ring["i"]
                ring["i"] = (i + 1) % len(ring["bufs"])
                if ring["work"][i] is not None:
                    ring["work"][i].wait()
                    ring["work"][i] = None
                fx = ring["bufs"][i]
                if ent["call"](fx, int(feed[self.step]) if self.rnnprop else 1):
                    self.last_path = "fused"
                    if self.sharded:
                        import torch.distributed as dist
                        ring["work"][i] = dist.all_reduce(fx, async_op=True)
                    return fx, [self.x[0].value]
                self._fast_unrolls.pop(fast_key, None)      # stale (the engine's workspace / layout changed): general path
        # restart = list of device tensors x0: run this unroll from x0 and the zero LSTM state / moments on the SAME
        # problem instance (rewind(x0) + launch); the fused kernels fold it in (no copy / memset pass), every other
        # path rewinds first
        restart_fused = False
        if restart is not None:
            if commit and record is None and len(self.x) == 1 and hasattr(eng, "mlp_unroll"):
                restart_fused = True                        # (decided for good below, once the path is known)
            else:
                self.rewind(restart)
        # placeholders
        # (persistent device buffers, re-uploaded only when a NEW array is fed: util.run_epoch feeds the same
        #  random scaling to every unroll of an epoch, DM/util.py:40-54; stable addresses keep plans valid)
        scales = []
        sbufs = self.__dict__.setdefault("_scale_bufs", {})
        for ph, var in zip(self.scale, self.x):
            if ph in feed:
                src = feed[ph]
                ent = sbufs.get(var.decl.name)
                if ent is None or ent[0] is not src:
                    arr = var._local(src)
                    B, D = self._panel_shape(var)
                    new = eng.tensor(arr.reshape(B, D))
                    if ent is not None and ent[1].shape == new.shape:
                        ent[1].copy_(new)
                        new = ent[1]
                    ent = sbufs[var.decl.name] = (src, new)
                scales.append(ent[1])
            else:
                scales.append(None)
        step0 = 1
        if self.rnnprop:
            if self.step not in feed:
                raise ValueError("You must feed a value for placeholder 'step' (DM/util.py:59-60)")
            step0 = int(feed[self.step])

        # buffers: run in place on the live tensors when `update` is fetched, on copies otherwise
        xs = [v.value if commit else v.value.clone() for v in self.x]
        slots = self.slots
        states, ms, vs = [], [], []
        for s in slots:
            if isinstance(s.state, PackedState):
                states.append(s.state if commit else s.state.clone())
            else:
                states.append(s.state)
            ms.append(s.m if (commit or s.m is None) else s.m.clone())
            vs.append(s.v if (commit or s.v is None) else s.v.clone())

        descs = []
        for v, sc in zip(self.x, scales):
            if self.term_of[v.decl.name].kind == _abi.PROB_MLP:
                descs.append(None)                          # (its x-scale is applied around l2o_mlp_fg, see _run_steps)
            else:
                descs.append(self._desc(v, sc))
        panels = []
        for xv, var in zip(xs, self.x):
            B, D = self._panel_shape(var)
            panels.append(xv.view(B, D))
        # the MLP optimizee is evaluated at x * scale and its gradient is scale * grad (DM/meta_dm_train.py:384)
        self._mlp_scales = [sc if self.term_of[v.decl.name].kind == _abi.PROB_MLP else None
                            for v, sc in zip(self.x, scales)]

        # fx[0..T] of this launch.  Sharded runs all-reduce it ASYNCHRONOUSLY (the next unroll
        # does not wait for the 404-byte collective); the buffers rotate so that a collective in
        # flight is never overwritten, and every reader goes through wait_fx().
        key = T
        ring = self._fx_cache.get(key)
        if ring is None:
            n = 4 if self.sharded else 1
            ring = self._fx_cache[key] = {"bufs": [eng.zeros(T + 1) for _ in range(n)], "work": [None] * n, "i": 0}
        fused_path = record is None and self._fused_ok(descs)
        if restart_fused and not fused_path:
            self.rewind(restart)                            # (panels / states are views of the live tensors: still valid)
            restart_fused = False
        if not fused_path:
            ring["i"] = 0                                  # a captured launch sequence owns buffer 0
        i = ring["i"]
        ring["i"] = (i + 1) % len(ring["bufs"]) if fused_path else 0
        if ring["work"][i] is not None:
            ring["work"][i].wait()
            ring["work"][i] = None
        fx = ring["bufs"][i]

        if events is not None:
            events[0].record()
        if record is not None and self.second_derivatives:
            record["x"] = []                              # the Hessian needs the iterates: step-granular recording
            record["descs"] = descs
        if record is not None and T > 0 and not self.second_derivatives and self._fused_ok(descs, record=True) \
                and isinstance(states[0], PackedState) and states[0].packed is not None:
            # meta-gradient on a fused-size problem: ONE launch that also records the history
            # (state before, gradient at, moments after every step; gradient at x_T)
            self.last_path = "fused"
            s, d = slots[0], descs[0]
            B, D = panels[0].shape
            N = B * D
            # the history buffers (and the per-step views of them, and the BPTT pointer table that
            # _bptt_panels keeps in this dict) live as long as the unroll keeps its shape
            key = (T, B, D, states[0].packed.numel(), bool(self.rnnprop))
            fp = self.__dict__.get("_fused_plan")
            if fp is None or fp["key"] != key:
                hist = {"st": eng.empty(T, states[0].packed.numel()), "g": eng.empty(T, N), "g_final": eng.empty(N)}
                if self.rnnprop:
                    hist.update(m=eng.empty(T, N), v=eng.empty(T, N))
                fp = self.__dict__["_fused_plan"] = dict(
                    key=key, hist=hist,
                    g=[[hist["g"][t].view(B, D)] for t in range(T)], st=[[hist["st"][t]] for t in range(T)],
                    m=[[hist["m"][t].view(B, D) if self.rnnprop else None] for t in range(T)],
                    v=[[hist["v"][t].view(B, D) if self.rnnprop else None] for t in range(T)],
                    g_final=[hist["g_final"].view(B, D)])
            hist = fp["hist"]
            fx_part = self._scratch("fx_part", (T + 1) * d.B_local)
            eng.unroll(s.net.spec, s.net.wpack(eng), d, panels[0], states[0].packed, ms[0], vs[0], T, step0,
                       fx_part, hist=hist, fx=fx)
            record.update(step0=step0, shapes=[tuple(pn.shape) for pn in panels], g=fp["g"], st=fp["st"],
                          m=fp["m"], v=fp["v"], g_final=fp["g_final"], plan=fp)
        elif (record is not None and not self.second_derivatives and not os.environ.get("L2O_NO_MLP_UNROLL_RECORD")
              and self._mlp_unroll_ok(slots, states, scales) >= (1 if os.environ.get("L2O_MLP_UNROLL_RECORD_GENERIC") else 2)):
            # meta-gradient on the neural optimizee: the T steps AND their history in ONE persistent launch
            # (l2o_mlp_unroll_record) instead of 3 T + 2 step-granular launches -- where the kernel's FAST form applies
            # (the reference's shape; training step 0.86 -> 0.68 ms at T = 20): its generic loops are no faster than
            # the step-granular path (minibatch 128: 1.12 vs 0.89 ms)
            self.last_path = "mlp_unroll"
            self._draw_minibatches(T)
            record.update(step0=step0, shapes=[tuple(pn.shape) for pn in panels])
            self._run_mlp_unroll_record(T, step0, panels, slots, states, ms, vs, scales, fx, record)
            if events is not None:
                events[1].record()
        elif record is not None:                           # meta-gradient: needs the per-step history
            self.last_path = "steps"
            self._draw_minibatches(T)
            record.update(step0=step0, shapes=[tuple(pn.shape) for pn in panels])
            self._run_steps(T, step0, descs, panels, slots, states, ms, vs, fx, record=record)
        elif fused_path:
            self.last_path = "fused"
            s, d = slots[0], descs[0]
            fx_part = self._scratch("fx_part", (T + 1) * d.B_local)
            x0v = restart[0].view(panels[0].shape) if restart_fused else None
            eng.unroll(s.net.spec, s.net.wpack(eng), d, panels[0], states[0].packed, ms[0], vs[0], T, step0,
                       fx_part, fx=fx, x0=x0v, zero_state=restart_fused)   # (the batch-mean reduction rides in the epilogue)
            if fast_key is not None and (restart is None or restart_fused) and d.x_scale is None:
                fu = self._fast_unrolls
                if len(fu) >= 16:
                    fu.pop(next(iter(fu)))                  # (oldest first: a bounded set of pinned buffers)
                call = eng.prepared_unroll(s.net.spec, s.net.wpack(eng), d, panels[0], states[0].packed, ms[0], vs[0],
                                           T, fx_part, x0v, restart_fused)
                if call is not None and getattr(s.net, "_wpack", None) is fast_objs[5]:
                    fu[fast_key] = {"call": call, "keep": fast_objs}
            if events is not None:
                events[1].record()
        elif self._mlp_unroll_ok(slots, states, scales):
            # the neural optimizee, all four variables stepped by one LSTM net: T steps in ONE persistent launch
            self.last_path = "mlp_unroll"
            self._draw_minibatches(T)
            term = self.terms[0]
            index_of = {v.decl.name: j for j, v in enumerate(self.x)}
            js = [index_of[tv.name] for tv in _term_vars(term)]
            slot_of = {s.var_index: si for si, s in enumerate(slots)}
            sis = [slot_of[j] for j in js]
            net = slots[sis[0]].net
            eng.mlp_unroll(net.spec, net.wpack(eng), self._mlp_desc(term), self._mlp_idx[0],
                           [panels[j] for j in js], [states[si].packed for si in sis], [ms[si] for si in sis],
                           [vs[si] for si in sis], [scales[j] for j in js], T, step0, fx)
            if events is not None:
                events[1].record()
        else:
            self.last_path = "steps"
            self._draw_minibatches(T)                      # host RNG -> persistent device index buffers
            graphable = (use_graph and commit and all(sc is None for sc in scales) and
                         all(isinstance(st, PackedState) for st in states) and hasattr(torch.cuda, "CUDAGraph")
                         and eng.device.type == "cuda")
            if not graphable:
                self._run_steps(T, step0, descs, panels, slots, states, ms, vs, fx)
            else:
                cache = self.__dict__.setdefault("_hip_graphs", {})
                entry = cache.get(step0)
                if entry is None:                          # 1st call: eager (allocates every scratch buffer)
                    self._run_steps(T, step0, descs, panels, slots, states, ms, vs, fx)
                    cache[step0] = "warm"
                else:
                    if entry == "warm":                    # 2nd call: capture (records, does not execute)
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g):
                            self._run_steps(T, step0, descs, panels, slots, states, ms, vs, fx)
                        cache[step0] = entry = g
                    entry.replay()
            if events is not None:
                events[1].record()

        if self.sharded:
            import torch.distributed as dist
            ring["work"][i] = dist.all_reduce(fx, async_op=True)
        if commit:
            for s, st in zip(slots, states):
                s.state = st
        return fx, xs

    def _run_mlp_unroll_record(self, T, step0, panels, slots, states, ms, vs, scales, fx, record):
        """The recording form of the fused MLP unroll: history buffers (built once per set of variable buffers, like the
        plan of _run_steps_planned, and handed to _backward in the same format) + one l2o_mlp_unroll_record launch."""
        eng = self.engine
        term = self.terms[0]
        nvar = len(self.x)
        index_of = {v.decl.name: j for j, v in enumerate(self.x)}
        js = [index_of[tv.name] for tv in _term_vars(term)]               # w1, b1, w2, b2 -> variable index
        slot_of = {s.var_index: si for si, s in enumerate(slots)}
        sis = [slot_of[j] for j in js]
        key = (T, tuple(p.data_ptr() for p in panels), tuple(st.packed.data_ptr() for st in states),
               tuple(0 if m is None else m.data_ptr() for m in ms))
        plan = self.__dict__.get("_mlp_record_plan")
        if plan is None or plan["key"] != key:
            hist_g = [eng.empty(T + 1, *panels[j].shape) for j in range(nvar)]
            hs = [eng.empty(max(T, 1), states[si].packed.numel()) for si in range(len(slots))]
            rn = ms[sis[0]] is not None
            hm = [eng.empty(T + 1, ms[si].numel()) if rn else None for si in range(len(slots))]
            hv = [eng.empty(T + 1, vs[si].numel()) if rn else None for si in range(len(slots))]
            plan = self.__dict__["_mlp_record_plan"] = dict(
                key=key, hist=dict(st=[hs[si] for si in sis], g=[hist_g[j] for j in js],
                                   m=[hm[si] for si in sis] if rn else None, v=[hv[si] for si in sis] if rn else None),
                g=[[hist_g[j][t] for j in range(nvar)] for t in range(T)],
                st=[[hs[si][t] for si in range(len(slots))] for t in range(T)],
                m=[[None if hm[si] is None else hm[si][t + 1] for si in range(len(slots))] for t in range(T)],
                v=[[None if hv[si] is None else hv[si][t + 1] for si in range(len(slots))] for t in range(T)],
                g_final=[hist_g[j][T] for j in range(nvar)])
        net = slots[sis[0]].net
        eng.mlp_unroll(net.spec, net.wpack(eng), self._mlp_desc(term), self._mlp_idx[0],
                       [panels[j] for j in js], [states[si].packed for si in sis], [ms[si] for si in sis],
                       [vs[si] for si in sis], [scales[j] for j in js], T, step0, fx, hist=plan["hist"])
        record.update(g=plan["g"], st=plan["st"], m=plan["m"], v=plan["v"], g_final=plan["g_final"], plan=plan)

    def _mlp_unroll_ok(self, slots, states, scales):
        """l2o_mlp_unroll applies: ONE problems.mnist term of weight 1 whose four variables are all stepped by the
        same (20, 20) LSTM net, on an engine / device that has the fused kernel."""
        eng = self.engine
        if not hasattr(eng, "mlp_unroll") or os.environ.get("L2O_DISABLE_FUSED") or self.sharded:
            return 0
        if len(self.terms) != 1 or self.terms[0].kind != _abi.PROB_MLP or self.terms[0].weight != 1.0:
            return 0
        tv = _term_vars(self.terms[0])
        if len(tv) != 4 or len(self.x) != 4 or len(slots) != 4:
            return 0
        net = slots[0].net
        for s, st in zip(slots, states):
            if s.net is not net or not isinstance(net, networks.StandardDeepLSTM) or not isinstance(st, PackedState) \
                    or st.packed is None:
                return 0
        return int(eng.mlp_unroll_supported(net.spec, self._mlp_desc(self.terms[0])))   # 2: the kernel's FAST form

    def wait_fx(self):
        """Make the current stream (NCCL) / the host (gloo) wait for the loss all-reduces in flight."""
        for ring in self._fx_cache.values():
            for k, w in enumerate(ring["work"]):
                if w is not None:
                    w.wait()
                    ring["work"][k] = None

    def _draw_minibatches(self, T):
        """A fresh uniform minibatch per evaluation of a neural optimizee (DM/problems.py:282-286: tf.random_uniform
        indices -- a device op there): indices [T+1, batch] in a PERSISTENT device buffer (so that a captured launch
        sequence sees the new indices).  Drawn ON THE DEVICE when the engine can (HipEngine.sample_int: a torch
        generator seeded from the stream of set_random_seed -- no host draw, no pageable upload that waits for the
        previous unroll); a `sampler` of the problem (parity tests) or L2O_HOST_SAMPLING=1: the host draw + upload."""
        bufs = self.__dict__.setdefault("_mlp_idx", {})
        eng = self.engine
        for k, term in enumerate(self.terms):
            if term.kind != _abi.PROB_MLP:
                continue
            d = self._mlp_desc(term)
            sampler = term.hyper.get("sampler")
            shape = (T + 1, d.batch)
            if sampler is None and hasattr(eng, "sample_int") and not os.environ.get("L2O_HOST_SAMPLING"):
                if k not in bufs or tuple(bufs[k].shape) != shape:
                    bufs[k] = eng.empty_int(*shape)
                eng.sample_int(bufs[k], d.images.shape[0], int(_rng.integers(0, 2 ** 62)))
                continue
            if sampler is None:
                idx = _rng.integers(0, d.images.shape[0], size=shape)
            else:
                idx = np.asarray(sampler(T + 1, d.batch, d.images.shape[0]))
            new = eng.int_tensor(idx.reshape(shape))
            if k in bufs and bufs[k].shape == new.shape:
                bufs[k].copy_(new)
            else:
                bufs[k] = new

    # -- meta-gradient (DM/meta.py:398-414) --------------------------------------------
    def train_step(self, feed, commit, learning_rate, defer=False):
        """One forward unroll (step-granular kernels, history recorded), back-propagation
        through time of loss = sum_t fx_t w.r.t. the optimizer networks' weights with the
        optimizee gradients held constant (tf.stop_gradient, DM/meta.py:328-329), and one Adam
        update of those weights (tf.train.AdamOptimizer(learning_rate).minimize(loss)).
        Returns the same dict as execute().  defer (Session.run(_defer_loss=True)): enqueue only -- no host sync, the
        loss entries of the result are None; honoured when every network's meta-step runs on the device."""
        record = {}
        fx, xs = self.launch(feed, commit, record=record)
        eng = self.engine
        T = self.len_unroll
        grads = self._backward(T, record)                   # (launched before the host reads anything back)
        self.wait_fx()
        fused = self.last_path in ("fused", "mlp_unroll") and hasattr(eng, "check_unroll_status")   # kernels with a status word
        # The meta-step goes out BEHIND the unroll and its back-propagation, before the host waits for the loss: with
        # every network on the device-side Adam the GPU then never idles while the host assembles the update (63 us of
        # a 0.40 ms step at config-2 size, T = 20).  A partner timeout of a fused unroll leaves a garbage history: the
        # device-side update is GUARDED by that unroll's status word (l2o_adam_step_guarded) and does not run; the host
        # learns of it at the sync below, takes the Adam step count back and raises.
        early = all(self._device_adam(self.nets[k]) for k in grads)
        if early:
            self._adam_apply(grads, learning_rate, guarded=fused)
        pend = self.__dict__.setdefault("_guarded_pending", 0)
        if defer and early:
            # nothing is read back: a failed unroll's status word is sticky, so the guarded updates of this and of every
            # later deferred step stay off until a synchronous step checks it, takes the step counts back and raises
            self._guarded_pending = pend + (1 if fused else 0)
            return {"loss": None, "fx": None, "fx_array": None,
                    "x": _LazyHost(eng, xs, [self._local_shape(var) for var in self.x])}
        if fused and hasattr(eng, "prefetch_unroll_status"):
            eng.prefetch_unroll_status()                    # (rides on the sync below)
        fx_host = eng.to_numpy(fx)                          # host sync
        if fused:
            self._guarded_pending = pend + (1 if early else 0)
            self._check_unroll_status()                     # raise BEFORE a host-side Adam update / report the skipped ones
        self._guarded_pending = 0
        x_out = _LazyHost(eng, xs, [self._local_shape(var) for var in self.x])   # copied to the host only if fetched
        if not early:
            self._adam_apply(grads, learning_rate)
        return {"loss": np.float32(fx_host.sum(dtype=np.float32)), "fx": np.float32(fx_host[T]),
                "x": x_out, "fx_array": fx_host}

    def _check_unroll_status(self):
        """engine.check_unroll_status() after a host sync; when it raises, guarded meta-steps enqueued since the last
        check did not run on the device (l2o_adam_step_guarded): the Adam step counts of ALL of them are taken back (those
        enqueued before the failing unroll did run -- the count errs on the low side).  The exception is FATAL for the
        optimizer state: Adam's step count no longer matches m / v, and in a sharded run only the failing rank
        skipped its update -- a caller that wants to continue must `restore` the last checkpoint (the training drivers do
        not catch it)."""
        try:
            self.engine.check_unroll_status()
        except Exception:
            n = self.__dict__.get("_guarded_pending", 0)
            if n and "_adam" in self.__dict__:
                self.__dict__["_adam"]["t"] -= n
            self._guarded_pending = 0
            raise

    def _bptt(self, net, acc, B, D, T, step0, gs, sts, ms, vs, dxs):
        """Back-propagation through T recorded steps of ONE network on one [B, D] panel:
        gs[t] the step's input gradient, sts[t] the packed state before it, ms / vs the RNNProp
        moments after it, dxs[t] = dL/d(delta_t).  Adds the weight gradients into ``acc``
        ({(module, variable): device tensor})."""
        self._bptt_panels(net, acc, T, step0, [dict(B=B, D=D, gs=gs, sts=sts, ms=ms, vs=vs, dxs=dxs)])

    def _bptt_panels(self, net, acc, T, step0, panels, cache=None):
        """The same for several panels (variables) that share the network: ONE backward launch
        per step for all of them (l2o_cwlstm_bwd_multi) when every panel is tile-aligned
        (D % 16 == 0 or B == 1), else panel by panel."""
        eng = self.engine
        b1, b2 = float(np.float32(self.beta1)), float(np.float32(self.beta2))
        spec = net.spec
        nl = len(spec.layers)
        fc = spec.preprocess == _abi.PRE_FC_ELU
        P = 20 if fc else (2 if spec.preprocess == _abi.PRE_LOGSIGN else 1)
        wdev = net.device_weights(eng)

        def add(mod, var, val):
            k = (mod, var)
            acc[k] = val if k not in acc else acc[k] + val

        second = any(pn.get("second") for pn in panels)

        def hess_update(pn, t, N):
            """second_derivatives: lam_t = g_t + lam_{t+1} + (d g_t / d x_t) u_t with u_t = dL/dg_t just emitted.
            g_t was recorded with its term's weight folded in (_run_steps), so the Hessian-vector product of the
            UNWEIGHTED optimizee carries the same factor."""
            hv = pn.setdefault("hv", eng.empty(N))
            eng.problem_hvp(pn["desc"], pn["xs"][t], pn["dg"].view(pn["B"], pn["D"]), hv.view(pn["B"], pn["D"]))
            w = float(pn.get("weight", 1.0))
            pn["lam"] = pn["gs"][t].reshape(N) + pn["lam"] + (hv if w == 1.0 else hv * w)   # (a new tensor: hv is reused)

        def rnnprop_input_adjoint(pn, t, N, Bt, k):
            """second_derivatives for RNNProp (DM/meta_rnnprop_train.py:380-388 without the stop_gradient): the network
            inputs m~ = m^/(sqrt(v^) + 1e-8), g~ = g/(sqrt(v^) + 1e-8) depend on g_t directly AND through the moment
            recurrences m_t = b1 m_{t-1} + (1 - b1) g_t, v_t = b2 v_{t-1} + (1 - b2) g_t^2 that later steps read.  From
            the step kernel's du (adjoint of the input projection's pre-activations) this forms u_t = dL/dg_t and the
            adjoints carried to step t - 1; elementwise device tensor code (a training-mode side path)."""
            du = Bt[:N, 8 * H + 1:8 * H + 1 + H]
            wfc = wdev["w_fc"].view(2, H)
            a0, a1 = (du * wfc[0]).sum(1), (du * wfc[1]).sum(1)                  # dL/dm~, dL/dg~
            g, m, v = pn["gs"][t].reshape(N), pn["ms"][t].reshape(N), pn["vs"][t].reshape(N)
            om1, om2 = 1.0 - b1 ** k, 1.0 - b2 ** k
            m_hat, sq = m / om1, torch.sqrt(v / om2)
            den = sq + 1e-8
            d_den = -(a0 * m_hat + a1 * g) / (den * den)
            d_vhat = torch.where(sq > 0, d_den * 0.5 / sq.clamp_min(1e-30), torch.zeros_like(sq))
            dm = a0 / den / om1 + pn["dm"]
            dv = d_vhat / om2 + pn["dv"]
            pn["dg"] = a1 / den + dm * float(np.float32(1.0 - self.beta1)) + dv * (2.0 * float(np.float32(1.0 - self.beta2))) * g
            pn["dm"], pn["dv"] = dm * b1, dv * b2

        def need_dxs():
            if second:                                      # running adjoint instead of the precomputed prefix sums
                for pn in panels:
                    N = pn["B"] * pn["D"]
                    pn["lam"] = pn["g_final"].reshape(N).clone()
                    pn["dg"] = eng.empty(N)
                    pn["dm"], pn["dv"] = eng.zeros(N), eng.zeros(N)      # RNNProp: adjoints of the carried moments
                return
            for pn in panels:                               # loss = sum_t fx_t: dL/d(delta_t) = g_final + sum_{tau > t} g_tau
                if pn.get("dxs") is None:
                    N = pn["B"] * pn["D"]
                    acc_g = pn["g_final"].reshape(N).clone()
                    pn["dxs"] = [None] * T
                    for t in reversed(range(T)):
                        pn["dxs"][t] = acc_g
                        acc_g = acc_g + pn["gs"][t].reshape(N)

        if spec.generic:
            # ANY `layers` tuple (DM/networks.py:157): the VALU backward companion of l2o_cwlstm_step_generic, one launch
            # per (step, panel); the weight gradients of a step are act_l^T dz_l per layer (l2o_atb), accumulated over
            # the steps.  A correct device path for the plugin contract, not a fast one.
            if second:
                raise NotImplementedError("second_derivatives=True is implemented for the (20, 20) and () nets")
            if not hasattr(eng, "bwd_step_generic"):
                raise NotImplementedError("this engine has no BPTT for layers=%r" % (spec.layers,))
            need_dxs()
            gen = net.wpack(eng)
            Hs = [int(h) for h in spec.layers]
            P = int(gen.c.in_dim)                           # (the fc width is the net's own, not the harness' 20)
            ins = [P] + Hs[:-1]
            for pn in panels:
                N = pn["B"] * pn["D"]
                nst = sum(2 * N * h for h in Hs)
                io = dict(act=[eng.empty(N, i + h) for i, h in zip(ins, Hs)], dz=[eng.empty(N, 4 * h) for h in Hs],
                          tc=eng.empty(N * sum(Hs)), h_last=eng.empty(N, Hs[-1]), dd=eng.empty(N))
                if fc:
                    io.update(feats=eng.empty(N, 2), du=eng.empty(N, P))
                carry_in, carry_out = eng.zeros(nst), eng.zeros(nst)
                for t in reversed(range(T)):
                    k = step0 + t
                    io.update(g=pn["gs"][t], m=pn["ms"][t], v=pn["vs"][t], st_prev=pn["sts"][t], dx_next=pn["dxs"][t],
                              carry_in=carry_in, carry_out=carry_out)
                    eng.bwd_step_generic(spec, gen, io, b1 ** k, b2 ** k, N)
                    carry_in, carry_out = carry_out, carry_in
                    def atb_blocks(A, Bmat):               # l2o_atb holds a KA <= 112 x KB <= 192 result in registers
                        ka, kb = A.shape[1], Bmat.shape[1]
                        if ka <= 112 and kb <= 192:
                            return eng.atb(A, Bmat)
                        rows = []
                        for r0 in range(0, ka, 96):
                            Ab = A[:, r0:r0 + 96].contiguous()
                            rows.append(torch.cat([eng.atb(Ab, Bmat[:, c0:c0 + 176].contiguous())
                                                   for c0 in range(0, kb, 176)], 1))
                        return torch.cat(rows, 0)
                    for l in range(nl):
                        add("lstm_%d" % (l + 1), "w_gates", atb_blocks(io["act"][l], io["dz"][l]))
                        add("lstm_%d" % (l + 1), "b_gates", io["dz"][l].sum(0))
                    dd = io["dd"].view(N, 1)
                    add("linear", "w", eng.atb(io["h_last"], dd))
                    add("linear", "b", dd.sum(0))
                    if fc:
                        add("input_projection", "w", eng.atb(io["feats"], io["du"]))
                        add("input_projection", "b", io["du"].sum(0))
            return
        if not nl:                                         # Linear-only net: two tiny products per step
            need_dxs()
            for pn in panels:
                N = pn["B"] * pn["D"]
                io = {"dd": eng.empty(N), "act1": eng.empty(N, 2)}
                for t in reversed(range(T)):
                    io.update(g=pn["gs"][t], dx_next=pn["lam"] if second else pn["dxs"][t], dg=pn.get("dg"))
                    eng.bwd_step(spec, wdev, io, b1 ** (step0 + t), b2 ** (step0 + t), pn["B"], pn["D"])
                    if second:
                        hess_update(pn, t, N)
                    dd = io["dd"].view(N, 1)
                    add("linear", "w", eng.atb(io["act1"], dd)[:P])
                    add("linear", "b", dd.sum(0))
            return
        # The kernel emits, per step and coordinate, one row of  A = [act1 | act2 | h2 | feats | 1]
        # and one of  Bm = [dz1 | dz2 | dd | du];  EVERY weight gradient of the unroll is a block of
        # the single product A^T Bm over all (step, panel, coordinate) rows.  (Three skinny rocBLAS
        # GEMMs per step cost 320 us; one chunked batched GEMM per unroll costs a few tens.)
        H = 20
        K1 = P + H
        KA = K1 + 2 * H + H + (2 if fc else 0) + 1
        KB = 4 * H + 4 * H + 1 + (H if fc else 0)
        multi = all(pn["D"] % 16 == 0 or pn["B"] == 1 for pn in panels) and len(panels) <= 8 and not second
        # the T-step launch takes ANY D (per-problem tiles with a ragged last one, the forward's packed-state layout);
        # the step-granular multi-panel kernel needs tile-aligned panels
        fused = (len(panels) <= 8 and not second and wdev.get("wpack") is not None
                 and not os.environ.get("L2O_BWD_STEPWISE") and hasattr(eng, "bwd_unroll")
                 and (multi or (getattr(eng, "bwd_unroll_any_d", False) and not os.environ.get("L2O_BWD_ALIGNED_ONLY")))
                 and _abi.get_option(_abi.OPT_BWD_KERNEL) == 0)                                   # A/B switches of the tests
        groups = [panels] if (multi or fused) else [[pn] for pn in panels]
        if not fused:
            need_dxs()
        for grp in groups:
            Ns = [pn["B"] * pn["D"] for pn in grp]
            if fused:                                       # rows = 16 x (B x ceil(D / 16)) per panel
                rows = [pn["B"] * ((pn["D"] + 15) // 16) * 16 for pn in grp]
            else:
                rows = [(n + 15) // 16 * 16 for n in Ns]
            offs = np.concatenate([[0], np.cumsum(rows)]).astype(int)   # row blocks (whole tiles)
            R = int(offs[-1])
            ragged = any(n % 16 for n in Ns)
            if fused:                                       # all T steps in one launch, the carries in registers;
                A, Bm = eng.empty(T, R, KA), eng.empty(T, R, KB)   # the kernel writes every row (padding rows as zeros)
                tkey = ("bwd_table", id(net), T)
                table = None if cache is None else cache.get(tkey)
                if table is None:
                    table = eng.bwd_table(grp, T) if hasattr(eng, "bwd_table") else None
                    if cache is not None:
                        cache[tkey] = table
                eng.bwd_unroll(spec, wdev, grp, T, step0, A, Bm, table=table)
            else:
                A = (eng.zeros if ragged else eng.empty)(T, R, KA)     # zero padding rows add nothing to A^T Bm
                Bm = (eng.zeros if ragged else eng.empty)(T, R, KB)
                for o, n in zip(offs[:-1], Ns):
                    A[:, o:o + n, KA - 1] = 1.0
                carry_in, carry_out = eng.zeros(4, R, H), eng.zeros(4, R, H)
            for t in (() if fused else reversed(range(T))):
                k = step0 + t
                At, Bt = A[t], Bm[t]
                if multi:
                    segs = [dict(g=pn["gs"][t], m=pn["ms"][t], v=pn["vs"][t], st_prev=pn["sts"][t], dx_next=pn["dxs"][t],
                                 B=pn["B"], D=pn["D"]) for pn in grp]
                    eng.bwd_multi(spec, wdev, segs, carry_in, carry_out, At, Bt, b1 ** k, b2 ** k)
                else:
                    pn, N = grp[0], Ns[0]
                    io = dict(g=pn["gs"][t], dx_next=pn["lam"] if second else pn["dxs"][t],
                              dg=None if fc else pn.get("dg"),     # (RNNProp: formed from du below, not by the kernel)
                              st_prev=pn["sts"][t], carry_in=carry_in[:, :N],
                              carry_out=carry_out[:, :N], m=pn["ms"][t], v=pn["vs"][t], a_stride=KA, b_stride=KB,
                              act1=At[:N, 0:K1], act2=At[:N, K1:K1 + 2 * H], h2=At[:N, K1 + 2 * H:K1 + 3 * H],
                              dz1=Bt[:N, 0:4 * H], dz2=Bt[:N, 4 * H:8 * H], dd=Bt[:N, 8 * H:8 * H + 1])
                    if fc:
                        io.update(feats=At[:N, K1 + 3 * H:K1 + 3 * H + 2], du=Bt[:N, 8 * H + 1:8 * H + 1 + H])
                    if R != N:                             # the generic kernel wants dense [4][N][H] carries
                        io["carry_in"], io["carry_out"] = carry_in[:, :N].contiguous(), eng.empty(4, N, H)
                    eng.bwd_step(spec, wdev, io, b1 ** k, b2 ** k, pn["B"], pn["D"])
                    if R != N:
                        carry_out[:, :N] = io["carry_out"]
                    if second:
                        if fc:
                            rnnprop_input_adjoint(pn, t, N, Bt, k)
                        hess_update(pn, t, N)
                carry_in, carry_out = carry_out, carry_in
            # l2o_cwlstm_wgrad: every weight gradient is a block of A^T Bm (only those blocks are computed)
            Gm = eng.wgrad(spec, A.view(T * R, KA), Bm.view(T * R, KB))
            blocks = [("lstm_1", "w_gates", 0, K1, 0, 4 * H), ("lstm_1", "b_gates", KA - 1, KA, 0, 4 * H),
                      ("lstm_2", "w_gates", K1, K1 + 2 * H, 4 * H, 8 * H), ("lstm_2", "b_gates", KA - 1, KA, 4 * H, 8 * H),
                      ("linear", "w", K1 + 2 * H, K1 + 3 * H, 8 * H, 8 * H + 1), ("linear", "b", KA - 1, KA, 8 * H, 8 * H + 1)]
            if fc:
                blocks += [("input_projection", "w", K1 + 3 * H, K1 + 3 * H + 2, 8 * H + 1, 8 * H + 1 + H),
                           ("input_projection", "b", KA - 1, KA, 8 * H + 1, 8 * H + 1 + H)]
            # (when this is the network's only contraction the meta-step reads the blocks in place: _adam_apply_device)
            only = len(groups) == 1 and not acc
            for mod, var, r0, r1, c0, c1 in blocks:
                blk = Gm[r0:r1, c0:c1] if var != "b_gates" and var != "b" else Gm[r0, c0:c1]
                add(mod, var, blk)
            srcs = self.__dict__.setdefault("_gm_src", {})
            if only:
                srcs[id(acc)] = (Gm, KB, {(mod, var): (r0, c0) for mod, var, r0, r1, c0, c1 in blocks})
            else:
                srcs.pop(id(acc), None)

    def _backward(self, T, rec):
        eng = self.engine
        step0 = rec["step0"]
        out = {}                                           # net key -> {(module, variable): device grad}
        by_net = {}                                        # variables that share a network go through ONE launch per step
        for si, s in enumerate(self.slots):
            net = s.net
            if not isinstance(net, networks.StandardDeepLSTM):
                continue
            j = s.var_index
            B, D = rec["shapes"][j]
            N = B * D
            # loss = sum_t fx_t and x_{t+1} = x_t + delta_t  =>  dL/d(delta_t) = sum_{tau > t} g_tau
            # (accumulated inside the fused BPTT kernel, or by _bptt_panels for the step-wise kernels)
            pn = dict(B=B, D=D, gs=[g[j] for g in rec["g"]], sts=[st[si] for st in rec["st"]],
                      ms=[m[si] for m in rec["m"]], vs=[v[si] for v in rec["v"]], dxs=None,
                      g_final=rec["g_final"][j].reshape(N))
            if self.second_derivatives:                    # dL/dx_t picks up H(x_t) . dL/dg_t (DM/meta.py:328-329)
                if rec["descs"][j] is None:
                    raise NotImplementedError("second_derivatives=True is implemented for the analytic optimizees")
                pn.update(second=True, desc=rec["descs"][j], xs=[x[j] for x in rec["x"]],
                          weight=self.term_of[self.x[j].decl.name].weight)
            by_net.setdefault(s.key, (net, []))[1].append(pn)
        for key, (net, panels) in by_net.items():      # rec["plan"]: buffers of a planned unroll are reused, so is the table
            self._bptt_panels(net, out.setdefault(key, {}), T, step0, panels, cache=rec.get("plan"))
        if self.sharded:
            # sum of the shards' weight gradients (1/B_global is already in every gradient): ONE collective
            # per network on a contiguous buffer -- the entries of `acc` are column blocks of A^T Bm, i.e.
            # NON-contiguous views, which RCCL rejects and gloo silently mis-reduces
            import torch.distributed as dist
            for acc in out.values():
                keys = sorted(acc)
                flat = torch.cat([acc[k].reshape(-1) for k in keys])
                dist.all_reduce(flat)
                off = 0
                for k in keys:
                    n = acc[k].numel()
                    acc[k] = flat[off:off + n].view(acc[k].shape)
                    off += n
        srcs = self.__dict__.get("_gm_src", {})
        if all(self._device_adam(self.nets[key]) for key in out):
            # the meta-step runs on the device: the gradients never visit the host
            res = {}
            for key, acc in out.items():
                src = None if self.sharded else srcs.pop(id(acc), None)
                res[key] = {k: _DevGrad(v, None if src is None else (src[0], src[1]) + src[2][k]) for k, v in acc.items()}
            return res
        srcs.clear()
        # ONE device-to-host copy for all weight gradients (each .cpu() is a stream sync + a transfer)
        items = [(key, k, v) for key, acc in out.items() for k, v in acc.items()]
        if not items:
            return {}
        flat = eng.to_numpy(torch.cat([v.reshape(-1) for _, _, v in items]))
        res, off = {}, 0
        for key, k, v in items:
            n = v.numel()
            res.setdefault(key, {})[k] = flat[off:off + n].reshape(tuple(v.shape))
            off += n
        return res

    def _device_adam(self, net):
        """Adam + weight re-pack on the device (l2o_adam_step, l2o_wpack_device) for the LSTM nets when the
        engine has them; L2O_HOST_ADAM=1 keeps the NumPy meta-step."""
        return (hasattr(self.engine, "adam_step") and isinstance(net, networks.StandardDeepLSTM)
                and len(net.spec.layers) > 0 and not net.spec.generic      # (generic `layers`: the host meta-step)
                and not os.environ.get("L2O_HOST_ADAM"))

    def _adam_apply_device(self, key, acc, st, lr_t, beta1, beta2, epsilon, guarded=False):
        """One network's meta-step without a host round trip: the gradients are laid out like the flat
        Sonnet-layout weight buffer (one torch.cat), l2o_adam_step updates that buffer in place and
        l2o_wpack_device rebuilds the MFMA-fragment copy from it.  The host dict goes stale (lazy refresh)."""
        import torch
        eng, net = self.engine, self.nets[key]
        wdev = net.device_weights(eng)
        buf, offs, names = net._wdev_buf, net._wdev_offs, net._wdev_names
        ds = st.setdefault("dev", {})
        ent = ds.get(key)
        if ent is None or ent["g"].numel() != buf.numel():
            ent = ds[key] = {"g": eng.zeros(buf.numel()), "m": eng.zeros(buf.numel()), "v": eng.zeros(buf.numel()),
                             "zeros": eng.zeros(8)}
        # every gradient a block of ONE contraction result (the usual case: one l2o_cwlstm_wgrad per network): the update
        # reads them in place through a static index map -- no slicing copies, no concatenation, no zero fills
        srcs = [getattr(acc.get(names[k]), "src", None) for k in offs]
        if (hasattr(eng, "adam_step_gather") and all(sr is not None for sr in srcs)
                and all(sr[0] is srcs[0][0] for sr in srcs) and not os.environ.get("L2O_NO_ADAM_GATHER")):
            G, KB = srcs[0][0], srcs[0][1]
            mkey = (KB, tuple((k, o, tuple(shp), srcs[i][2], srcs[i][3]) for i, (k, (o, shp)) in enumerate(offs.items())))
            gmap = ent.get("gmap")
            if gmap is None or gmap[0] != mkey:
                idx = np.full(buf.numel(), -1, np.int32)
                for i, (k, (o, shp)) in enumerate(offs.items()):
                    r0, c0 = srcs[i][2], srcs[i][3]
                    n = int(np.prod(shp))
                    cols = int(shp[-1]) if len(shp) > 1 else n            # a bias is ONE row of G
                    e = np.arange(n)
                    idx[o:o + n] = (r0 + (e // cols if len(shp) > 1 else 0)) * KB + c0 + e % cols
                gmap = ent["gmap"] = (mkey, eng.int_tensor(idx))
            eng.adam_step_gather(buf, ent["m"], ent["v"], G, gmap[1], lr_t, beta1, beta2, epsilon, guarded=guarded)
            eng.pack_weights_device(net.spec, wdev, wdev["wpack"])
            net.mark_device_updated()
            return
        parts, pos = [], 0
        for k, (o, shp) in offs.items():                   # buffer order; 16-byte aligned parts
            n = int(np.prod(shp))
            if o > pos:
                parts.append(ent["zeros"][:o - pos])
            gk = acc.get(names[k])
            if gk is None:
                parts.append(torch.zeros(n, dtype=torch.float32, device=buf.device))
            else:
                t = gk.t if isinstance(gk, _DevGrad) else eng.tensor(np.asarray(gk, np.float32))
                parts.append(t.reshape(-1))
            pos = o + n
        if buf.numel() > pos:
            parts.append(ent["zeros"][:buf.numel() - pos])
        torch.cat(parts, out=ent["g"])
        if guarded:
            eng.adam_step(buf, ent["m"], ent["v"], ent["g"], lr_t, beta1, beta2, epsilon, guarded=True)
        else:
            eng.adam_step(buf, ent["m"], ent["v"], ent["g"], lr_t, beta1, beta2, epsilon)
        eng.pack_weights_device(net.spec, wdev, wdev["wpack"])
        net.mark_device_updated()

    def _adam_apply(self, grads, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8, slot="_adam", guarded=False):
        """tf.train.AdamOptimizer's update (TF 1.x `_apply_dense`): lr_t = lr sqrt(1-b2^t)/(1-b1^t);
        m <- b1 m + (1-b1) g; v <- b2 v + (1-b2) g^2; var <- var - lr_t m / (sqrt(v) + eps).
        A few thousand weights: done on the host in fp32, then re-packed for the kernels."""
        st = self.__dict__.setdefault(slot, {"t": 0, "m": {}, "v": {}})   # one tf.train.AdamOptimizer per slot
        st["t"] += 1
        t = st["t"]
        f = np.float32
        lr_t = f(learning_rate * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
        for key, acc in grads.items():                     # one flat vector per network: a handful of NumPy calls
            net = self.nets[key]
            if self._device_adam(net):
                self._adam_apply_device(key, acc, st, lr_t, beta1, beta2, epsilon, guarded=guarded)
                continue
            names = list(acc.keys())
            g = np.concatenate([np.asarray(acc[k], np.float32).reshape(-1) for k in names])
            w = np.concatenate([net.variables[mod][var].reshape(-1) for mod, var in names])
            kk = (key, tuple(names))
            m = st["m"].get(kk)
            if m is None:
                m, v = np.zeros_like(g), np.zeros_like(g)
            else:
                v = st["v"][kk]
            m = f(beta1) * m + f(1.0 - beta1) * g
            v = f(beta2) * v + f(1.0 - beta2) * g * g
            st["m"][kk], st["v"][kk] = m, v
            w = w - lr_t * m / (np.sqrt(v) + f(epsilon))
            off = 0
            for mod, var in names:
                n = net.variables[mod][var].size
                net.assign(mod, var, w[off:off + n])
                off += n

    def gradients(self, feed=None):
        """[d f(x * scale) / d x_j] at the current variables as device tensors (panel shaped),
        without touching any state -- what DM/data_generator.py:44-45 builds with tf.gradients."""
        T = self.len_unroll
        self.len_unroll = 0
        feed = dict(feed or {})
        if self.rnnprop:
            feed.setdefault(self.step, 1)                   # no optimizer step is taken: any value does
        try:
            rec = {}
            self.launch(feed, commit=False, record=rec)
        finally:
            self.len_unroll = T
        return rec["g_final"]

    def rewind(self, x0):
        """Device-side restart of the SAME problem instance: x <- x0 (list of device tensors),
        LSTM state / moments <- 0, without re-sampling the problem data (bench.py)."""
        self._ensure_init()
        dst, src = [v.value for v in self.x], list(x0)
        zeros = self.__dict__.setdefault("_rewind_zeros", {})
        for s in self.slots:
            ts = []
            if isinstance(s.state, PackedState) and s.state.packed is not None:
                ts.append(s.state.packed)
            if s.m is not None:
                ts += [s.m, s.v]
            for t in ts:
                z = zeros.get(tuple(t.shape))
                if z is None or z.device != t.device:
                    z = zeros[tuple(t.shape)] = torch.zeros_like(t)
                dst.append(t)
                src.append(z)
        if hasattr(torch, "_foreach_copy_") and len(dst) > 1:
            torch._foreach_copy_(dst, src)                  # ONE launch for x, LSTM state and moments
        else:
            for d_, s_ in zip(dst, src):
                d_.copy_(s_)

    def _local_shape(self, var):
        if self.sharded:
            return (self.shard[1] - self.shard[0],) + tuple(var.shape[1:])
        return var.shape

    def _scratch(self, name, n):
        buf = getattr(self, "_scr_" + name, None)
        if buf is None or buf.numel() < n:
            buf = self.engine.zeros(n)
            setattr(self, "_scr_" + name, buf)
        return buf[:n]

    def _mlp_desc(self, term):
        """Device copy of the dataset of a problems.mnist term (uploaded once)."""
        cache = self.__dict__.setdefault("_mlp_cache", {})
        key = id(term.hyper["images"])
        if key not in cache:
            from ._engine import MlpDesc
            images = np.ascontiguousarray(term.hyper["images"], np.float32).reshape(len(term.hyper["labels"]), -1)
            w1 = term.var[0]
            cache[key] = MlpDesc(n_in=images.shape[1], n_hidden=w1.shape[1], n_out=term.var[2].shape[1],
                                 batch=int(term.hyper["batch_size"]),
                                 activation=0 if term.hyper["activation"] == "sigmoid" else 1,
                                 images=self.engine.tensor(images),
                                 labels=self.engine.int_tensor(term.hyper["labels"]))
        return cache[key]

    def _run_steps(self, T, step0, descs, panels, slots, states, ms, vs, fx, record=None):
        """Step-granular path: per step one forward+gradient launch per loss term
        (l2o_problem_fg / l2o_mlp_fg) and one l2o_cwlstm_step per (net, variable)."""
        eng = self.engine
        nvar = len(self.x)
        index_of = {v.decl.name: j for j, v in enumerate(self.x)}
        if record is not None and self._plan_ok(slots, states, nvar):
            return self._run_steps_planned(T, step0, panels, slots, states, ms, vs, fx, record, index_of)
        grads = [self._scratch("g%d" % j, panels[j].numel()).view(panels[j].shape) for j in range(nvar)]
        tmp = self._scratch("fx1", 1)
        single = len(self.terms) == 1 and self.terms[0].weight == 1.0
        b1, b2 = float(np.float32(self.beta1)), float(np.float32(self.beta2))
        mlp_idx = self.__dict__.get("_mlp_idx", {})
        # one analytic term of weight 1: the per-problem losses of all T+1 steps are kept and
        # reduced over the batch by ONE launch at the end (like the fused path) instead of a
        # tiny reduction kernel per step
        defer = single and self.terms[0].kind != _abi.PROB_MLP
        if defer:
            jd = index_of[self.terms[0].var.name]
            f_all = self._scratch("f_all", (T + 1) * descs[jd].B_local)

        def forward(t, want_grad):
            if not single:
                fx[t:t + 1].zero_()
            for k, term in enumerate(self.terms):
                out = fx[t:t + 1] if single else tmp
                if term.kind == _abi.PROB_MLP:
                    js = [index_of[tv.name] for tv in _term_vars(term)]
                    sc = getattr(self, "_mlp_scales", None) or [None] * nvar
                    xin = [panels[j] if sc[j] is None else
                           torch.mul(panels[j], sc[j], out=self._scratch("xs%d" % j, panels[j].numel()).view(panels[j].shape))
                           for j in js]
                    eng.mlp_fg(self._mlp_desc(term), mlp_idx[k][t], *xin, out,
                               [grads[j] for j in js] if want_grad else None)
                    if want_grad:
                        for j in js:
                            if sc[j] is not None:
                                grads[j].mul_(sc[j])
                else:
                    js = [index_of[term.var.name]]
                    j = js[0]
                    if defer:
                        Bl = descs[j].B_local
                        eng.problem_fg(descs[j], panels[j], f_all[t * Bl:(t + 1) * Bl], grads[j] if want_grad else None)
                        continue
                    f_part = self._scratch("f%d" % j, descs[j].B_local)
                    eng.problem_fg(descs[j], panels[j], f_part, grads[j] if want_grad else None)
                    eng.reduce_fx(f_part, 1, descs[j].B_local, descs[j].B_global, out)
                if not single:
                    fx[t:t + 1].add_(tmp, alpha=float(term.weight))
                    if want_grad and term.weight != 1.0:
                        for j in js:
                            grads[j].mul_(float(term.weight))

        chain = None
        if record is not None:
            record.update(g=[], st=[], m=[], v=[])
            # History without copies: the gradients are written straight into their [T + 1] history
            # buffers, and the LSTM state / RNNProp moments are CHAINED through [T + 1] buffers --
            # step t reads slice t and writes slice t + 1 (l2o_step_seg.st_out / m_out / v_out).
            hist_g = [eng.empty(T + 1, *panels[j].shape) for j in range(nvar)]
            chain = {}
            for si, s in enumerate(slots):
                if isinstance(s.net, networks.StandardDeepLSTM) and states[si].packed is not None:
                    hs = eng.empty(T + 1, states[si].packed.numel())
                    hs[0].copy_(states[si].packed)
                    hm = hv = None
                    if ms[si] is not None:
                        hm, hv = eng.empty(T + 1, ms[si].numel()), eng.empty(T + 1, vs[si].numel())
                        hm[0].copy_(ms[si].reshape(-1)); hv[0].copy_(vs[si].reshape(-1))
                    chain[si] = (hs, hm, hv)
        for t in range(T):
            if record is not None:
                grads[:] = [hg[t] for hg in hist_g]
            forward(t, True)
            k = step0 + t
            if record is not None:
                if "x" in record:
                    record["x"].append([pn.clone() for pn in panels])
                record["g"].append(list(grads))
                record["st"].append([chain[si][0][t] if si in chain else
                                     (None if not isinstance(st, PackedState) or st.packed is None else st.packed.clone())
                                     for si, st in enumerate(states)])
            # variables that share a network are updated by ONE launch (DM/meta.py:330-336 applies
            # `net` to every variable of its subset inside the same time step)
            groups = {}
            for si, s in enumerate(slots):
                j = s.var_index
                B, D = panels[j].shape
                if isinstance(s.net, networks.StandardDeepLSTM):
                    if chain is not None and si in chain:
                        hs, hm, hv = chain[si]
                        seg = (grads[j], None if hm is None else hm[t].view(B, D), None if hv is None else hv[t].view(B, D),
                               hs[t], panels[j], B, D, hs[t + 1], None if hm is None else hm[t + 1].view(B, D),
                               None if hv is None else hv[t + 1].view(B, D))
                    else:
                        seg = (grads[j], ms[si], vs[si], None if states[si].packed is None else states[si].packed,
                               panels[j], B, D)
                    groups.setdefault(id(s.net), (s.net, []))[1].append(seg)
                else:                                    # Sgd / Adam baseline nets
                    delta, states[si] = s.net(grads[j], states[si])
                    panels[j].add_(delta.view(B, D))
            for net, segs in groups.values():
                eng.lstm_step_multi(net.spec, net.wpack(eng), segs, b1 ** k, b2 ** k)
            if record is not None:                         # RNNProp moments AFTER this step's update
                record["m"].append([(chain[si][1][t + 1] if si in chain else mm.clone()) if mm is not None else None
                                    for si, mm in enumerate(ms)])
                record["v"].append([(chain[si][2][t + 1] if si in chain else vv.clone()) if vv is not None else None
                                    for si, vv in enumerate(vs)])
        if record is not None:
            grads[:] = [hg[T] for hg in hist_g]
        forward(T, record is not None)                     # training also needs the gradient at x_T
        if defer:
            eng.reduce_fx(f_all, T + 1, descs[jd].B_local, descs[jd].B_global, fx)
        if record is not None:
            record["g_final"] = list(grads)
            for si, (hs, hm, hv) in chain.items():         # the variables take the end of the chain
                states[si].packed.copy_(hs[T])
                if hm is not None:
                    ms[si].copy_(hm[T].view(ms[si].shape)); vs[si].copy_(hv[T].view(vs[si].shape))


    # -- the recorded unroll of a neural optimizee as a PLAN: buffers and ctypes arguments built once ------
    def _plan_ok(self, slots, states, nvar):
        eng = self.engine
        if os.environ.get("L2O_NO_STEP_PLAN") or not hasattr(eng, "prepared_mlp_fg"):
            return False
        if len(self.terms) != 1 or self.terms[0].kind != _abi.PROB_MLP or self.terms[0].weight != 1.0:
            return False
        if len(_term_vars(self.terms[0])) != nvar:
            return False
        per_net = collections.Counter()
        for s, st in zip(slots, states):
            if not isinstance(s.net, networks.StandardDeepLSTM) or not isinstance(st, PackedState) or st.packed is None:
                return False
            per_net[id(s.net)] += 1
        return all(n <= eng.MAX_STEP_SEGS for n in per_net.values())

    def _run_steps_planned(self, T, step0, panels, slots, states, ms, vs, fx, record, index_of):
        """_run_steps(record=...) for ONE MLP loss term whose variables are all updated by LSTM nets: the
        [T + 1] history buffers (gradients written in place, state / moments chained through them) and
        the ctypes argument objects of the 2T + 1 launches are built once and reused by every unroll
        with the same buffers; per step the host only passes what changes (the loss slot, the
        bias-correction powers, the address of the re-packed weights)."""
        eng = self.engine
        term = self.terms[0]
        nvar = len(self.x)
        idxbuf = self._mlp_idx[0]
        sc = getattr(self, "_mlp_scales", None) or [None] * nvar
        scaled = any(x is not None for x in sc)
        key = (T, idxbuf.data_ptr(), tuple(p.data_ptr() for p in panels),
               tuple(st.packed.data_ptr() for st in states), tuple(0 if m is None else m.data_ptr() for m in ms),
               tuple(0 if x is None else x.data_ptr() for x in sc))
        plan = self.__dict__.get("_step_plan")
        if plan is None or plan["key"] != key:
            js = [index_of[tv.name] for tv in _term_vars(term)]
            hist_g = [eng.empty(T + 1, *panels[j].shape) for j in range(nvar)]
            chain = []
            for si, s in enumerate(slots):
                hs = eng.empty(T + 1, states[si].packed.numel())
                hm = hv = None
                if ms[si] is not None:
                    hm, hv = eng.empty(T + 1, ms[si].numel()), eng.empty(T + 1, vs[si].numel())
                chain.append((hs, hm, hv))
            desc = self._mlp_desc(term)
            # x-scale (random rescaling of the optimizee, DM/util.py:40-54): evaluate at xs = x * scale (three
            # multi-tensor launches per step: copy, multiply, and scale the gradients afterwards)
            ones = None
            xs_in = [panels[j] for j in js]
            if scaled:
                ones = [x if x is not None else torch.ones_like(panels[j]) for j, x in enumerate(sc)]
                xs_in = [eng.empty(*panels[j].shape) for j in js]
            mlp = [eng.prepared_mlp_fg(desc, idxbuf[t], *xs_in, [hist_g[j][t] for j in js])
                   for t in range(T + 1)]
            lstm = []
            for t in range(T):
                groups = {}
                for si, s in enumerate(slots):
                    j = s.var_index
                    B, D = panels[j].shape
                    hs, hm, hv = chain[si]
                    seg = (hist_g[j][t], None if hm is None else hm[t].view(B, D), None if hv is None else hv[t].view(B, D),
                           hs[t], panels[j], B, D, hs[t + 1], None if hm is None else hm[t + 1].view(B, D),
                           None if hv is None else hv[t + 1].view(B, D))
                    groups.setdefault(id(s.net), (s.net, []))[1].append(seg)
                lstm.append([(net, eng.prepared_lstm_step_multi(net.spec, segs)) for net, segs in groups.values()])
            plan = self.__dict__["_step_plan"] = dict(
                key=key, chain=chain, mlp=mlp, lstm=lstm, xs_in=xs_in if scaled else None,
                x_src=[panels[j] for j in js], sc=[ones[j] for j in js] if scaled else None,
                g_steps=[[hist_g[j][t] for j in js] for t in range(T + 1)],
                g=[[hist_g[j][t] for j in range(nvar)] for t in range(T)],
                st=[[chain[si][0][t] for si in range(len(slots))] for t in range(T)],
                m=[[None if chain[si][1] is None else chain[si][1][t + 1] for si in range(len(slots))] for t in range(T)],
                v=[[None if chain[si][2] is None else chain[si][2][t + 1] for si in range(len(slots))] for t in range(T)],
                g_final=[hist_g[j][T] for j in range(nvar)])
        chain = plan["chain"]
        # the variables' state / moments enter slot 0 of the history chain and leave from slot T: ONE multi-tensor copy
        # each way (12 + 12 single copies per training step on the four MLP variables before)
        heads, tails, cur = [], [], []
        for si, (hs, hm, hv) in enumerate(chain):
            heads.append(hs[0]); tails.append(hs[T]); cur.append(states[si].packed)
            if hm is not None:
                heads += [hm[0], hv[0]]; tails += [hm[T], hv[T]]
                cur += [ms[si].view(-1), vs[si].view(-1)]
        torch._foreach_copy_(heads, cur)
        import ctypes
        wp = {}
        for calls in plan["lstm"][:1]:
            for net, _ in calls:
                wp[id(net)] = (net.wpack(eng), ctypes.c_void_p(net.wpack(eng).data_ptr()))
        b1, b2 = float(np.float32(self.beta1)), float(np.float32(self.beta2))
        fxp = fx.data_ptr()
        mlp, lstm = plan["mlp"], plan["lstm"]
        xs_in, x_src, scl, g_steps = plan["xs_in"], plan["x_src"], plan["sc"], plan["g_steps"]
        for t in range(T + 1):                             # (training also needs the gradient at x_T)
            if xs_in is not None:
                torch._foreach_copy_(xs_in, x_src)
                torch._foreach_mul_(xs_in, scl)
            mlp[t](fxp + 4 * t)
            if xs_in is not None:
                torch._foreach_mul_(g_steps[t], scl)
            if t == T:
                break
            k = step0 + t
            p1, p2 = b1 ** k, b2 ** k
            for net, call in lstm[t]:
                call(wp[id(net)][1], p1, p2)
        record.update(g=plan["g"], st=plan["st"], m=plan["m"], v=plan["v"], g_final=plan["g_final"], plan=plan)
        torch._foreach_copy_(cur, tails)                   # the variables take the end of the chain


# ---------------------------------------------------------------------------
# Imitation ("multi-task") unrolls of the train forks
# ---------------------------------------------------------------------------
class MtUnroll(object):
    """One imitation-learning unroll: the optimizer networks are fed a recorded gradient
    sequence of an analytic optimizer (``mt_inputs``, [T, P] per subset) and regress its updates
    (``mt_labels``):  loss_mt = sum_t 0.5 sum_s ||label_t - net(input_t, state_t)||^2 / P_total,
    with its own LSTM state (and RNNProp moments) carried between unrolls by ``update_mt``.
    DM/meta_dm_train.py:421-499, 515-523; DM/meta_rnnprop_train.py:437-555, 567-584.

    Forward = T launches of l2o_cwlstm_step on a zeroed scratch iterate (x <- 0 + delta), the
    Adam step back-propagates through them with l2o_cwlstm_bwd_step (dL/d(delta_t) =
    (delta_t - label_t) / P_total).  ``Session.run`` drives it like an UnrollGraph
    (reset / execute / train_step)."""

    def __init__(self, graph, mti):
        self.graph, self.mti = graph, mti
        self.len_unroll = graph.len_unroll
        self.learning_rate = 0.01
        T = self.len_unroll
        self.keys = list(graph.net_keys)
        self.sizes = [int(sum(int(np.prod(graph.x[j].shape)) for j in subset)) for subset in graph.subsets]
        self.total = int(sum(self.sizes))
        for k in self.keys:
            if not isinstance(graph.nets[k], networks.StandardDeepLSTM):
                raise NotImplementedError("imitation unrolls are implemented for the LSTM optimizer networks")
        self.labels = [Placeholder("mt%d_label_subset%d" % (mti, j), (T, P)) for j, P in enumerate(self.sizes)]
        self.inputs = [Placeholder("mt%d_input_subset%d" % (mti, j), (T, P)) for j, P in enumerate(self.sizes)]
        self.state = None

    @property
    def engine(self):
        return self.graph.engine

    def reset(self):
        eng, g = self.engine, self.graph
        self.state, self.m, self.v = [], [], []
        for k, P in zip(self.keys, self.sizes):
            self.state.append(PackedState.zeros(eng, 1, P, g.nets[k].spec.layers))
            self.m.append(eng.zeros(1, P) if g.rnnprop else None)
            self.v.append(eng.zeros(1, P) if g.rnnprop else None)

    def _forward(self, feed, commit, record=None):
        eng, g = self.engine, self.graph
        if self.state is None:
            self.reset()
        T = self.len_unroll
        feed = feed or {}
        for ph in self.inputs + self.labels:
            if ph not in feed:
                raise ValueError("You must feed a value for placeholder %r" % (ph.name,))
        step0 = 1
        if g.rnnprop:
            if g.step not in feed:
                raise ValueError("You must feed a value for placeholder 'step' (DM/util.py:59-60)")
            step0 = int(feed[g.step])
        b1, b2 = float(np.float32(g.beta1)), float(np.float32(g.beta2))
        states = [st if commit else st.clone() for st in self.state]
        ms = [m if (commit or m is None) else m.clone() for m in self.m]
        vs = [v if (commit or v is None) else v.clone() for v in self.v]
        loss = eng.zeros(1)
        inv = 1.0 / float(self.total)
        if record is not None:
            record.update(step0=step0, g=[], st=[], m=[], v=[], dx=[])
        ins = [eng.tensor(np.asarray(feed[ph], np.float32).reshape(T, P)) for ph, P in zip(self.inputs, self.sizes)]
        labs = [eng.tensor(np.asarray(feed[ph], np.float32).reshape(T, P)) for ph, P in zip(self.labels, self.sizes)]
        for t in range(T):
            k = step0 + t
            rg, rst, rm, rv, rdx = [], [], [], [], []
            for si, (key, P) in enumerate(zip(self.keys, self.sizes)):
                net = g.nets[key]
                gin = ins[si][t].view(1, P)
                delta = eng.zeros(1, P)
                if record is not None:
                    rg.append(gin)
                    rst.append(None if states[si].packed is None else states[si].packed.clone())
                eng.lstm_step(net.spec, net.wpack(eng), gin, ms[si], vs[si], b1 ** k, b2 ** k,
                              None if states[si].packed is None else states[si].packed, delta, 1, P)
                diff = delta.view(P) - labs[si][t]
                loss += (0.5 * inv) * (diff * diff).sum()
                if record is not None:
                    rm.append(None if ms[si] is None else ms[si].clone())
                    rv.append(None if vs[si] is None else vs[si].clone())
                    rdx.append(diff * inv)
            if record is not None:
                for lst, val in zip((record["g"], record["st"], record["m"], record["v"], record["dx"]),
                                    (rg, rst, rm, rv, rdx)):
                    lst.append(val)
        return loss

    def execute(self, feed, commit):
        loss = self._forward(feed, commit)
        return {"loss": np.float32(self.engine.to_numpy(loss)[0])}

    def train_step(self, feed, commit, learning_rate, defer=False):
        """loss_mt + one step of this task's own tf.train.AdamOptimizer (DM/meta_dm_train.py:549-553)."""
        rec = {}
        loss = self._forward(feed, commit, record=rec)
        g = self.graph
        T = self.len_unroll
        out = {}
        for si, (key, P) in enumerate(zip(self.keys, self.sizes)):
            g._bptt(g.nets[key], out.setdefault(key, {}), 1, P, T, rec["step0"], [r[si] for r in rec["g"]],
                    [r[si] for r in rec["st"]], [r[si] for r in rec["m"]], [r[si] for r in rec["v"]],
                    [r[si] for r in rec["dx"]])
        eng = self.engine
        if all(g._device_adam(g.nets[key]) for key in out):
            grads = {key: {k: _DevGrad(v) for k, v in acc.items()} for key, acc in out.items()}
        else:
            grads = {key: {k: eng.to_numpy(v) for k, v in acc.items()} for key, acc in out.items()}
        g._adam_apply(grads, learning_rate, slot="_adam_mt%d" % self.mti)
        return {"loss": np.float32(eng.to_numpy(loss)[0])}


def make_mt_handles(graph, num_mt):
    """(loss_mt, steps_mt, update_mt, reset_mt, mt_labels, mt_inputs) of the train forks."""
    if not hasattr(graph, "mt") or len(graph.mt) != num_mt:
        graph.mt = [MtUnroll(graph, i) for i in range(num_mt)]
    loss_mt = [Fetch(m, "loss", "loss_mt%d" % i) for i, m in enumerate(graph.mt)]
    steps_mt = [Fetch(m, "step", "step_mt%d" % i) for i, m in enumerate(graph.mt)]
    update_mt = [[Fetch(m, "update", "update_mt%d" % i)] for i, m in enumerate(graph.mt)]
    reset_mt = [[Fetch(m, "reset", "reset_mt%d" % i)] for i, m in enumerate(graph.mt)]
    return loss_mt, steps_mt, update_mt, reset_mt, [m.labels for m in graph.mt], [m.inputs for m in graph.mt]


# ---------------------------------------------------------------------------
# MetaOptimizer
# ---------------------------------------------------------------------------
class MetaOptimizer(object):
    """Learning to learn (meta) optimizer.  DM/meta.py:219-414.

    Optimizer which has an internal RNN which takes as input, at each iteration,
    the gradient of the function being minimized and returns a step direction.
    """

    _rnnprop = False

    def __init__(self, **kwargs):
        """``**kwargs`` maps network identifiers to ``networks.factory`` parameters
        (DM/meta.py:228-253); no kwargs = the default coordinate-wise LogAndSign net."""
        self._nets = None
        self._graph = None
        self.beta1 = self.beta2 = 0.95
        if not kwargs:
            self._config = {k: dict(v) for k, v in _DEFAULT_CONFIG.items()}
        else:
            self._config = kwargs

    @property
    def graph(self):
        """The UnrollGraph of the last meta_loss call (variables, placeholders, launch())."""
        return self._graph

    # -- checkpoints: DM/meta.py:255-267, DM/meta_dm_train.py:257-302 ----------
    def save(self, sess=None, path=None, index=None):
        """Save meta-optimizer: ``{path}/{k}.l2l`` (or ``.l2l-{index}``), dill pickles of
        {module: {variable: ndarray}}."""
        result = {}
        for k, net in self._nets.items():
            if path is None:
                filename = None
                key = k
            elif index is not None:
                filename = os.path.join(path, "{}.l2l-{}".format(k, index))
                key = filename
            else:
                filename = os.path.join(path, "{}.l2l".format(k))
                key = filename
            net_vars = networks.save(net, sess, filename=filename)
            result[key] = net_vars
        return result

    def restorer(self):
        """DM/meta_dm_train.py:274-287 builds assign placeholders; nothing to build here."""

    def restore(self, sess, path, index):
        """DM/meta_dm_train.py:289-302: load ``{k}.l2l-{index}`` back into the live nets."""
        import dill as pickle
        for k, net in self._nets.items():
            filename = os.path.join(path, "{}.l2l-{}".format(k, index))
            with open(filename, "rb") as f:
                data = pickle.load(f)
            for module_name, variables in net.variables.items():
                for variable_name in variables:
                    net.assign(module_name, variable_name, data[module_name][variable_name])

    # -- the unroll ------------------------------------------------------------
    def _build_graph(self, make_loss, len_unroll, net_assignments, second_derivatives):
        graph = UnrollGraph(self, make_loss, len_unroll, net_assignments, rnnprop=self._rnnprop,
                            beta1=self.beta1, beta2=self.beta2)
        # DM/meta.py:328-329: without the flag the optimizee gradients are constants of the meta-gradient
        # (tf.stop_gradient); with it dL/dx_t also receives H(x_t) . dL/dg_t (l2o_problem_hvp)
        graph.second_derivatives = bool(second_derivatives)
        self._graph = graph
        return graph

    @staticmethod
    def _handles(graph):
        return MetaLoss(Fetch(graph, "loss"), [Fetch(graph, "update")], [Fetch(graph, "reset")],
                        Fetch(graph, "fx"), [Fetch(graph, ("x", j), "x_final_%d" % j) for j in range(len(graph.x))])

    def meta_loss(self, make_loss, len_unroll, net_assignments=None, second_derivatives=False):
        """Returns handles computing the meta-loss: namedtuple (loss, update, reset, fx, x).
        DM/meta.py:269-396."""
        return self._handles(self._build_graph(make_loss, len_unroll, net_assignments, second_derivatives))

    def meta_minimize(self, make_loss, len_unroll, learning_rate=0.01, **kwargs):
        """Returns handles minimizing the meta-loss: namedtuple (step, update, reset, fx, x).
        DM/meta.py:398-414: ``step`` = one Adam update of the optimizer networks on the
        meta-loss of the unroll (truncated BPTT: x and the LSTM state are carried over by
        ``update`` without gradient, the optimizee gradients are constants)."""
        info = self.meta_loss(make_loss, len_unroll, **kwargs)
        self._graph.learning_rate = learning_rate
        return MetaStep(Fetch(self._graph, "step"), *info[1:])
