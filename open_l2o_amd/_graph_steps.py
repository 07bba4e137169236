"""UnrollGraph's step-granular and neural-optimizee execution plans (split out of meta.py in round 4): l2o_problem_fg /
l2o_mlp_fg + l2o_cwlstm_step per step (eager, planned with prepared calls, or recorded for BPTT), the minibatch draws of
problems.mnist, and the persistent MLP unroll's recording form."""
from __future__ import annotations

import collections  # noqa: F401
import os

import numpy as np
import torch

from . import _abi, networks
from ._graph_core import PackedState, _DevGrad, _LazyHost, _term_vars, _world, rng  # noqa: F401


class StepPlanMixin(object):
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
        if sampler is None and hasattr(eng, "sample_int"):
            idx = eng.empty_int(n, L + 1, d.batch)           # drawn on the device, like _draw_minibatches (one call)
            eng.sample_int(idx, d.images.shape[0], int(rng().integers(0, 2 ** 62)))
        else:
            rows = []
            for _ in range(n):                               # the same draws, in the same order, as n x _draw_minibatches(L)
                if sampler is None:
                    rows.append(rng().integers(0, d.images.shape[0], size=(L + 1, d.batch)))
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

    def mlp_instance(self, feed=None, dry=False, draw=True):
        """What this graph contributes to a launch of several optimizee instances (replicas.Replicas ->
        l2o_mlp_unroll_multi): its minibatch indices (drawn here, like a launch of its own would), the live x / LSTM
        state / moment buffers of the four variables in the order w1, b1, w2, b2 and its loss buffer -- or None when
        the fused MLP unroll does not apply to it.  dry: no draw, no buffers -- only whether it applies.  draw=False: the
        caller has filled self._mlp_idx[0] itself (Replicas draws the minibatches of all its replicas in one call)."""
        self._ensure_init()
        T = self.len_unroll
        feed = feed or {}
        if any(ph in feed for ph in self.scale):
            return None
        slots = self.slots
        states = [s.state for s in slots]
        if not self._mlp_unroll_ok(slots, states, [None] * len(self.x)):
            return None
        term = self.terms[0]
        net = slots[0].net
        desc = self._mlp_desc(term)
        if dry:
            return dict(net=net, desc=desc)
        if self.rnnprop and self.step not in feed:
            raise ValueError("You must feed a value for placeholder 'step' (DM/util.py:59-60)")
        if draw:
            self._draw_minibatches(T)
        index_of = {v.decl.name: j for j, v in enumerate(self.x)}
        js = [index_of[tv.name] for tv in _term_vars(term)]               # w1, b1, w2, b2 -> variable index
        slot_of = {s.var_index: si for si, s in enumerate(slots)}
        sis = [slot_of[j] for j in js]
        panels = [v.value.view(*self._panel_shape(v)) for v in self.x]
        ring = self._fx_cache.get(T)
        if ring is None:
            store = self.engine.zeros(1, T + 1)
            ring = self._fx_cache[T] = {"store": store, "bufs": [store[0]], "work": [None], "i": 0, "pending": []}
        return dict(net=net, desc=desc, indices=self._mlp_idx[0], xs=[panels[j] for j in js],
                    sts=[states[si].packed for si in sis], ms=[slots[si].m for si in sis], vs=[slots[si].v for si in sis],
                    scales=[None] * 4, fx=ring["bufs"][0])

    def _draw_minibatches(self, T):
        """A fresh uniform minibatch per evaluation of a neural optimizee (DM/problems.py:282-286: tf.random_uniform
        indices -- a device op there): indices [T+1, batch] in a PERSISTENT device buffer (so that a captured launch
        sequence sees the new indices).  Drawn ON THE DEVICE when the engine can (HipEngine.sample_int: a torch
        generator seeded from the stream of set_random_seed -- no host draw, no pageable upload that waits for the
        previous unroll); a `sampler` of the problem (parity tests): the host draw + upload."""
        bufs = self.__dict__.setdefault("_mlp_idx", {})
        eng = self.engine
        if self.__dict__.get("_reuse_minibatches") and bufs:
            return                                          # (recovery re-run of an unroll: the minibatches it drew)
        for k, term in enumerate(self.terms):
            if term.kind != _abi.PROB_MLP:
                continue
            d = self._mlp_desc(term)
            sampler = term.hyper.get("sampler")
            shape = (T + 1, d.batch)
            if sampler is None and hasattr(eng, "sample_int"):
                if k not in bufs or tuple(bufs[k].shape) != shape:
                    bufs[k] = eng.empty_int(*shape)
                eng.sample_int(bufs[k], d.images.shape[0], int(rng().integers(0, 2 ** 62)))
                continue
            if sampler is None:
                idx = rng().integers(0, d.images.shape[0], size=shape)
            else:
                idx = np.asarray(sampler(T + 1, d.batch, d.images.shape[0]))
            new = eng.int_tensor(idx.reshape(shape))
            if k in bufs and bufs[k].shape == new.shape:
                bufs[k].copy_(new)
            else:
                bufs[k] = new

    def _mlp_desc(self, term):
        """Device copy of the dataset of a problems.mnist term (uploaded once)."""
        cache = self.__dict__.setdefault("_mlp_cache", {})
        key = id(term.hyper["images"])
        layers = tuple(term.hyper.get("layers") or (term.var[0].shape[1],))
        key = (key, layers)
        if key not in cache:
            from ._engine import MlpDeepDesc, MlpDesc
            images = np.ascontiguousarray(term.hyper["images"], np.float32).reshape(len(term.hyper["labels"]), -1)
            common = dict(batch=int(term.hyper["batch_size"]), activation=0 if term.hyper["activation"] == "sigmoid" else 1,
                          images=self.engine.tensor(images), labels=self.engine.int_tensor(term.hyper["labels"]))
            if len(layers) == 1:
                cache[key] = MlpDesc(n_in=images.shape[1], n_hidden=term.var[0].shape[1], n_out=term.var[2].shape[1], **common)
            else:                                           # "mnist_deeper": the step-granular kernels (l2o_mlp_deep_fg)
                cache[key] = MlpDeepDesc(n_in=images.shape[1], hidden=layers, n_out=term.var[-1].shape[0], **common)
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
                    if len(js) == 4:
                        eng.mlp_fg(self._mlp_desc(term), mlp_idx[k][t], *xin, out,
                                   [grads[j] for j in js] if want_grad else None)
                    else:                                   # several hidden layers
                        eng.mlp_deep_fg(self._mlp_desc(term), mlp_idx[k][t], xin, out,
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
        if len(_term_vars(self.terms[0])) != nvar or nvar != 4:    # (the prepared calls are l2o_mlp_fg's: one hidden layer)
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
