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
from ._graph_adam import AdamMixin
from ._graph_bptt import BpttMixin
from ._graph_core import (Fetch, Placeholder, Variable, PackedState, _make_nets, _DEFAULT_CONFIG, _DevGrad, _LazyHost,  # noqa: F401
                          _term_vars, _world, _all_reduce, synced_scale, local_slice, _Slot, set_random_seed, rng, _RngBox)
from ._graph_steps import StepPlanMixin

MetaLoss = collections.namedtuple("MetaLoss", "loss, update, reset, fx, x")     # DM/meta.py:158
MetaStep = collections.namedtuple("MetaStep", "step, update, reset, fx, x")     # DM/meta.py:159


# ---------------------------------------------------------------------------
# the unroll "graph": construction, reset, launch (this module); BPTT (_graph_bptt), the meta-step (_graph_adam), the
# step-granular / neural-optimizee plans (_graph_steps); handles and variables (_graph_core)
# ---------------------------------------------------------------------------
class UnrollGraph(BpttMixin, AdamMixin, StepPlanMixin, object):
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
                cache["src"], cache["val"] = wc, self.engine.colsum(wc.contiguous())      # (l2o_colsum, batched)
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
        fx_host, fx, xs = self._sync_or_recover(fx, xs, feed, commit)
        x_out = _LazyHost(eng, xs, [self._local_shape(var) for var in self.x])   # copied to the host only if fetched
        # (host NumPy on the T + 1 losses already copied back: loss = tf.reduce_sum(fx_array), DM/meta.py:376)
        return {"loss": np.float32(fx_host.sum(dtype=np.float32)), "fx": np.float32(fx_host[T]),
                "x": x_out, "fx_array": fx_host}

    # -- recovery from a partner timeout (round 5) -----------------------------------------
    # The two-CU unroll and the persistent MLP unroll exchange data between workgroups with bounded spins; a timeout (a
    # transient loss of co-residency: another process's kernels holding CUs) invalidates that launch's x_T / state, not its
    # inputs.  An EVALUATION unroll is re-run from what it started from on the exchange-free kernels of the same shapes
    # (k_unroll_lds / k_unroll: L2O_OPT_PAIR = 0; the step-granular MLP path: L2O_OPT_MLP_UNROLL = 0) and logged.  What it
    # started from is either the caller's x0 + the zero state (`restart=`: nothing to keep) or the live buffers, which an
    # in-place launch of an exchanging kernel snapshots first (one multi-tensor copy).  L2O_NO_RECOVERY=1: raise instead.
    def _snapshot(self, slots):
        """Copy x, packed LSTM states and moments aside ahead of an IN-PLACE launch that may time out."""
        live = [v.value for v in self.x]
        for s in slots:
            if isinstance(s.state, PackedState) and s.state.packed is not None:
                live.append(s.state.packed)
            if s.m is not None:
                live += [s.m, s.v]
        snap = self.__dict__.get("_snap")
        if snap is None or len(snap["bak"]) != len(live) or any(b.shape != t.shape for b, t in zip(snap["bak"], live)):
            snap = self._snap = {"bak": [torch.empty_like(t) for t in live]}
        if hasattr(torch, "_foreach_copy_") and len(live) > 1:
            torch._foreach_copy_(snap["bak"], live)
        else:
            for b, t in zip(snap["bak"], live):
                b.copy_(t)
        snap["live"] = live
        self._last_launch["snapshot"] = True

    def _wants_snapshot(self, restart_fused):
        """An in-place committed launch of a kernel that can time out: unknown before the first launch of a shape (then:
        yes), afterwards what the library reported for it (l2o_last_unroll_form)."""
        if restart_fused or os.environ.get("L2O_NO_RECOVERY") or not hasattr(self.engine, "last_unroll_exchanges"):
            return False
        if self.__dict__.get("_last_launch", {}).get("restart") is not None:
            return False                                 # (restart= launches are re-run from the caller's x0: nothing to keep)
        return self.__dict__.get("_exchanges", True)

    def _reduce_status(self):
        """Sharded graphs: MAX-reduce the sticky status word of this unroll over the ranks, on the device and ahead of the
        host sync, so that EVERY rank sees a partner timeout of ANY rank (a rank that recovers alone issues
        one loss all-reduce more than its peers, and the healthy ranks keep the contaminated fx).  The collective is
        unconditional on a sharded graph -- ranks whose launch has no workspace contribute 0 -- so it pairs up whatever
        kernel form each rank's shard selected.  Returns the reduced word (device int32 [1]) or None."""
        eng = self.engine
        if not (self.sharded and hasattr(eng, "unroll_status_tensor")):
            return None
        stw = eng.unroll_status_tensor() if self.last_path in ("fused", "mlp_unroll") else None
        red = self.__dict__.get("_status_red")
        if red is None:
            red = self._status_red = torch.zeros(1, dtype=torch.int32, device=eng.device)
        if stw is None:
            red.zero_()
        else:
            red.copy_(stw)
        _all_reduce(red, op="MAX")
        if stw is not None:
            stw.copy_(red)                               # (this rank's own check below then raises on a peer's timeout too)
        return red

    def _sync_or_recover(self, fx, xs, feed, commit):
        """The host sync of an evaluation unroll + the status check; on a partner timeout the unroll is re-run on the
        exchange-free kernels -- by every rank of a sharded job together.  Returns (fx on the host, fx, xs)."""
        eng = self.engine
        self.wait_fx()
        has_status = hasattr(eng, "check_unroll_status")
        fused = self.last_path in ("fused", "mlp_unroll") and has_status
        red = self._reduce_status() if has_status else None
        if fused and hasattr(eng, "prefetch_unroll_status"):
            eng.prefetch_unroll_status()                 # (rides on the sync below)
        fx_host = eng.to_numpy(fx)                       # host sync
        if not fused and red is None:
            return fx_host, fx, xs
        if fused and hasattr(eng, "last_unroll_exchanges"):
            self._exchanges = bool(eng.last_unroll_exchanges())
        try:
            if fused:
                self._check_unroll_status()
            if red is not None and int(red[0]):          # (a rank without a status word of its own: a peer timed out)
                raise _abi.L2OPartnerTimeout(_abi.L2O_ERR_TIMEOUT, "a peer rank's unroll reported a partner timeout")
        except _abi.L2OPartnerTimeout as err:
            info = self.__dict__.get("_last_launch", {})
            # a sticky status raised by ANOTHER graph's deferred training unroll on this engine is not ours to recover
            # from: that graph takes its Adam step counts back and the error surfaces
            other = getattr(eng, "_deferred_graph", None)
            other = other() if other is not None else None
            if other is not None and other is not self and other.__dict__.get("_guarded_pending", 0):
                other._take_back_pending()
                raise
            can = not os.environ.get("L2O_NO_RECOVERY") and (info.get("restart") is not None or info.get("snapshot")
                                                             or not info.get("commit", True))
            if red is not None:
                # every rank must take the same decision: recover only if ALL ranks hold what their unroll started from
                flag = torch.full((1,), 1 if can else 0, dtype=torch.int32, device=eng.device)
                _all_reduce(flag, op="MIN")
                can = bool(int(flag[0]))
            if not can:
                raise
            import warnings
            warnings.warn("open_l2o_amd: %s -- re-running this unroll on the exchange-free kernels" % (err,), RuntimeWarning)
            self.recoveries = self.__dict__.get("recoveries", 0) + 1
            if info.get("restart") is None and info.get("snapshot"):    # the in-place launch: its inputs come back
                snap = self._snap
                for t, b in zip(snap["live"], snap["bak"]):
                    t.copy_(b)
            self._reuse_minibatches = True               # (the MLP optimizee: the SAME minibatches, not a fresh draw)
            try:
                with _abi.option_scope({_abi.OPT_PAIR: 0, _abi.OPT_MLP_UNROLL: 0}):
                    fx, xs = self.launch(feed, commit, restart=info.get("restart"), _recovering=True)
                    self.wait_fx()
                    fx_host = eng.to_numpy(fx)
                    self._check_unroll_status()          # (an exchange-free kernel raises nothing)
            finally:
                self._reuse_minibatches = False
        return fx_host, fx, xs

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
            feed = {self.step: 1} if self.rnnprop else {}
            fx, xs = self.launch(feed, commit=True)
            fx_host, _, _ = self._sync_or_recover(fx, xs, feed, True)
        finally:
            self.len_unroll = L
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

    def launch(self, feed=None, commit=True, events=None, use_graph=False, record=None, restart=None, _recovering=False):
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
        if not _recovering:
            self._last_launch = {"restart": restart if (restart is not None and commit and record is None) else None,
                                 "snapshot": False, "commit": bool(commit)}
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
                if commit and restart is None and not _recovering and self._wants_snapshot(False):
                    self._snapshot(self.slots)
                ring = self._fx_cache[T]
                i = ring["i"]
                ring["i"] = (i + 1) % len(ring["bufs"])
                self._claim_fx(ring, i)
                fx = ring["bufs"][i]
                if ent["call"](fx, int(feed[self.step]) if self.rnnprop else 1):
                    self.last_path = "fused"
                    if self.sharded:
                        self._queue_fx(ring, i)
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
            n = self.FX_RING if self.sharded else 1
            store = eng.zeros(n, T + 1)                      # (ONE tensor: a run of pending buffers is one contiguous all-reduce)
            ring = self._fx_cache[key] = {"store": store, "bufs": [store[k] for k in range(n)], "work": [None] * n, "i": 0,
                                          "pending": []}
        fused_path = record is None and self._fused_ok(descs)
        if restart_fused and not fused_path:
            self.rewind(restart)                            # (panels / states are views of the live tensors: still valid)
            restart_fused = False
        if not fused_path:
            ring["i"] = 0                                  # a captured launch sequence owns buffer 0
        i = ring["i"]
        ring["i"] = (i + 1) % len(ring["bufs"]) if fused_path else 0
        self._claim_fx(ring, i)
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
            if commit and not _recovering and self._wants_snapshot(restart_fused):
                self._snapshot(slots)
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
            if commit and not _recovering and self._wants_snapshot(False):
                self._snapshot(slots)
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
            self._queue_fx(ring, i)
        if commit:
            for s, st in zip(slots, states):
                s.state = st
        return fx, xs

    # -- the loss all-reduce of a sharded job: DEFERRED and COALESCED (round 6) --------------------------------------------------
    # Nothing reads an unroll's losses before the host asks for them (Session.run: wait_fx + a copy; bench.py: at the end of the
    # timed region), so the T + 1 partial sums of an unroll stay in their ring buffer, marked pending, and ONE all-reduce over
    # the contiguous run of pending buffers goes out when somebody reads (wait_fx) or when the ring is about to wrap.  Why it
    # matters: the two-CU unroll needs every CU of the device at once, so a collective kernel running beside it on RCCL's
    # stream holds back the NEXT unroll's launch until it has finished -- measured with a world-size-1 RCCL group, one
    # all-reduce per unroll cost 11.8 us of a 188 us unroll (6.84 -> 6.43 G on config 4's shard of 8); a real 8-rank ring is
    # slower than that.  With one collective per 15 unrolls the same measurement still showed 6.8 us per unroll (a collective
    # costs the device ~100 us of hiccup whatever its size), hence a ring of 64: one collective per 63 unread unrolls.
    FX_RING = 64

    def _claim_fx(self, ring, i):
        """Buffer i is about to be overwritten: its pending / in-flight reduction must have gone out and finished."""
        if i in ring.get("pending", ()):
            self._flush_fx(ring)
        if ring["work"][i] is not None:
            ring["work"][i].wait()
            ring["work"][i] = None

    def _queue_fx(self, ring, i):
        ring["pending"].append(i)
        if len(ring["pending"]) >= len(ring["bufs"]) - 1:
            self._flush_fx(ring)

    def _flush_fx(self, ring):
        """One asynchronous all-reduce per contiguous run of pending loss buffers."""
        pend = sorted(ring.get("pending", ()))
        ring["pending"] = []
        k = 0
        while k < len(pend):
            e = k
            while e + 1 < len(pend) and pend[e + 1] == pend[e] + 1:
                e += 1
            w = _all_reduce(ring["store"][pend[k]:pend[e] + 1], async_op=True)
            for j in range(k, e + 1):
                ring["work"][pend[j]] = w
            k = e + 1

    def wait_fx(self):
        """Issue the pending loss all-reduces and make the current stream (NCCL) / the host (gloo) wait for them."""
        for ring in self._fx_cache.values():
            if ring.get("pending"):
                self._flush_fx(ring)
            done = set()
            for k, w in enumerate(ring["work"]):
                if w is not None:
                    if id(w) not in done:
                        w.wait()
                        done.add(id(w))
                    ring["work"][k] = None

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
        if self.sharded and hasattr(eng, "check_unroll_status"):
            # every rank must take the SAME decision about this meta-step: the status words of the ranks'
            # unrolls are MAX-reduced in place, on the device, ahead of the guarded update -- one rank's partner timeout
            # skips the update (and raises at the next host check) on all of them
            self._reduce_status()
        if early:
            self._adam_apply(grads, learning_rate, guarded=fused)
        pend = self.__dict__.setdefault("_guarded_pending", 0)
        if defer and early:
            # nothing is read back: a failed unroll's status word is sticky, so the guarded updates of this and of every
            # later deferred step stay off until a synchronous step checks it, takes the step counts back and raises
            self._guarded_pending = pend + (1 if fused else 0)
            if fused:
                import weakref
                eng._deferred_graph = weakref.ref(self)     # (whose sticky status it is, should an evaluation see it first)
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
        optimizer state of a TRAINING step: Adam's step count no longer matches m / v -- a caller that wants
        to continue must `restore` the last checkpoint (the training drivers do not catch it); in a sharded run the status
        word was MAX-reduced over the ranks ahead of the guarded update (train_step), so every rank skipped it and every
        rank raises here.  Evaluation unrolls recover instead (_sync_or_recover)."""
        try:
            self.engine.check_unroll_status()
        except Exception:
            self._take_back_pending()
            raise

    def _take_back_pending(self):
        n = self.__dict__.get("_guarded_pending", 0)
        if n and "_adam" in self.__dict__:
            self.__dict__["_adam"]["t"] -= n
        self._guarded_pending = 0

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
        dst, src = self.rewind_lists(x0)
        if hasattr(torch, "_foreach_copy_") and len(dst) > 1:
            torch._foreach_copy_(dst, src)                  # ONE launch for x, LSTM state and moments
        else:
            for d_, s_ in zip(dst, src):
                d_.copy_(s_)

    def rewind_lists(self, x0):
        """(destinations, sources) of rewind(x0): a caller that rewinds several graphs at once (bench.py's replicas) passes
        the concatenated lists to ONE multi-tensor copy."""
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
        return dst, src

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
                diff = eng.lincomb(eng.empty(P), delta.view(P), 1.0, labs[si][t], -1.0)      # delta - label
                ss = eng.atb(diff.view(P, 1), diff.view(P, 1))                                # |diff|^2 (l2o_atb, 1 x 1)
                eng.lincomb(loss, loss, 1.0, ss.view(1), 0.5 * inv)
                if record is not None:
                    rm.append(None if ms[si] is None else ms[si].clone())
                    rv.append(None if vs[si] is None else vs[si].clone())
                    rdx.append(eng.lincomb(eng.empty(P), diff, inv))
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
