"""Handles, variables and small helpers of the unroll graph (split out of meta.py in round 4: Fetch / Placeholder,
Variable, PackedState, net construction, device-gradient wrappers, rank helpers).  ``meta`` re-exports all of it."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _abi, networks


class _RngBox(object):
    """The process-wide NumPy generator behind ``set_random_seed`` (one object every module of the package shares; the
    attribute is replaced on re-seeding, so holders of the BOX never see a stale stream)."""
    gen = np.random.default_rng(0)


def rng():
    return _RngBox.gen


def set_random_seed(seed):
    """Seed for the optimizee initialisers (x0, W, y ...) and the network weights --
    the analogue of ``tf.set_random_seed`` (DM/evaluate_dm.py:51-52)."""
    _RngBox.gen = np.random.default_rng(seed)
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
            arr = rng().standard_normal(shape, dtype=np.float32) * np.float32(init[2]) + np.float32(init[1])
        elif init[0] == "uniform":
            arr = rng().random(shape, dtype=np.float32) * np.float32(init[2] - init[1]) + np.float32(init[1])
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
        seed and keeps its shard (the ranks together hold the problem batch a single process would)."""
        eng = self._graph.engine
        init = self.decl.initializer
        if init is not None and init[0] in ("normal", "uniform") and hasattr(eng, "sample"):
            seed = int(rng().integers(0, 2 ** 62))
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


_EMULATED_WORLD = None
_EMULATED_COLLECTIVES = False


def emulate_world(rank=None, world=None, collectives=False):
    """Run ONE shard of a sharded job in a single process: graphs built afterwards behave as rank ``rank`` of ``world``
    (contiguous batch slice, 1/B_global in every gradient); the losses they report are the SHARD's partial sums / B_global.
    For measuring the per-rank rate of a configuration that is defined on more GPUs than the box has (bench.py
    --emulate-world: BASELINE config 4's shard of 8).  collectives=False: no process group, the collectives are skipped.
    collectives=True (round 6): every collective of the data path is ISSUED through the initialised torch.distributed
    group of this process (a world-size-1 RCCL group on a 1-GPU box: the communicator, the stream hand-over and the
    enqueue path are the real ones, the result is unchanged) -- what a rank pays for the collective, measurable without a
    second GPU.  emulate_world() ends it."""
    global _EMULATED_WORLD, _EMULATED_COLLECTIVES
    if rank is None or world is None or int(world) <= 1:
        _EMULATED_WORLD, _EMULATED_COLLECTIVES = None, False
    else:
        if not 0 <= int(rank) < int(world):
            raise ValueError("emulate_world: rank %r outside world %r" % (rank, world))
        if collectives:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("emulate_world(collectives=True) needs an initialised torch.distributed process group")
        _EMULATED_WORLD, _EMULATED_COLLECTIVES = (int(rank), int(world)), bool(collectives)


def _world():
    if _EMULATED_WORLD is not None:
        return _EMULATED_WORLD
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _all_reduce(t, async_op=False, op=None):
    """torch.distributed.all_reduce(t) over the job's ranks -- the ONLY collective of the data path (the
    T + 1 partial losses per unroll; the flat weight gradient and the unroll status word per training step).  A no-op
    under emulate_world (one shard measured on its own) unless it was asked to keep the collectives."""
    if _EMULATED_WORLD is not None and not _EMULATED_COLLECTIVES:
        return None
    import torch.distributed as dist
    if op is None:
        return dist.all_reduce(t, async_op=async_op)
    return dist.all_reduce(t, op=getattr(dist.ReduceOp, op), async_op=async_op)


def synced_scale(shape, bound):
    """exp(U[-bound, bound]) of the given (GLOBAL) shape from ``np.random`` (DM/util.py:44), identical on
    every rank: drawn on rank 0 and broadcast when torch.distributed is initialised."""
    arr = np.exp(np.random.uniform(-bound, bound, size=tuple(shape)))
    if _world()[1] > 1 and _EMULATED_WORLD is None:
        import torch.distributed as dist
        box = [arr]
        dist.broadcast_object_list(box, src=0)
        arr = box[0]
    return arr


def local_slice(var, arr):
    """This rank's part of a global-shape array that belongs to ``var`` (the whole array when unsharded)."""
    return var._local(arr).reshape(var._graph._local_shape(var)) if isinstance(var, Variable) else np.asarray(arr)



class _Slot(object):
    """One (net, variable) pairing: the LSTM state (and RNNProp moments) of a variable."""

    def __init__(self, key, net, var_index):
        self.key, self.net, self.var_index = key, net, var_index
        self.state = None      # PackedState | Adam tuple | []
        self.m = None
        self.v = None


