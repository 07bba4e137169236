"""Several independent optimizee instances stepped by ONE optimizer network, unrolled TOGETHER (round 6).

The reference runs one unroll of one optimizee instance per ``sess.run`` (DM/meta_rnnprop_train.py:397-423,
DM/util.py:31-89); a meta-training batch of optimizees, or BASELINE config 5's replicas, are N such unrolls.  On the
MI355X the neural optimizee (problems.mnist, 784-20-10) has two kernel forms:

    form "chip"  k_mlp_unroll: ONE instance on all 8 XCDs -- the lowest latency of a single unroll
    form "xcd"   k_mlp_xcd:    one instance per XCD, up to 8 per launch -- 3-4 x the throughput

``Replicas`` builds N unroll graphs of one ``MetaOptimizer`` that share its networks and runs them in launches of up to
eight instances (l2o_mlp_unroll_multi), or one after the other on the whole chip:

    reps = Replicas(optimizer, [problems.mnist(...) for _ in range(8)], len_unroll=200)
    reps.reset()
    fx = reps.run({step: 1})            # -> [8] final losses; reps.fx_arrays: the T + 1 losses of every instance

Every instance draws its own initial weights and its own minibatches, exactly as N separate ``meta_loss`` graphs would.
"""
from __future__ import annotations

import os
import warnings

import numpy as np

from . import _abi
from ._graph_core import _term_vars


class Replicas(object):
    def __init__(self, optimizer, make_losses, len_unroll, net_assignments=None):
        if not make_losses:
            raise ValueError("Replicas needs at least one problem")
        self.optimizer = optimizer
        self.graphs = []
        for make_loss in make_losses:
            g = optimizer._build_graph(make_loss, len_unroll, net_assignments, False)
            if self.graphs:                                  # ONE set of networks for all instances (the first graph's)
                g0 = self.graphs[0]
                if sorted(g.nets) != sorted(g0.nets):
                    raise ValueError("the replicas disagree on the optimizer's networks")
                g.nets = g0.nets
                for s in g.slots:
                    s.net = g0.nets[s.key]
                g._mlp_cache = g0.__dict__.setdefault("_mlp_cache", {})    # (one device copy of the data set)
            self.graphs.append(g)
        optimizer._graph, optimizer._nets = self.graphs[0], self.graphs[0].nets
        self.len_unroll = int(len_unroll)
        self.last_form = None
        self.recoveries = 0
        self.fx_arrays = None

    @property
    def step(self):
        """RNNProp's `step` placeholder (the same object for every replica's feed)."""
        return self.graphs[0].step

    def reset(self):
        for g in self.graphs:
            g.reset()

    def _feed(self, g, feed):
        """`feed` is written against the FIRST graph's placeholders; every replica gets the same values."""
        if not feed:
            return {}
        g0 = self.graphs[0]
        out = {}
        for ph, val in feed.items():
            if ph is g0.step:
                out[g.step] = val
            elif ph in g0.scale:
                raise ValueError("x-scale placeholders are not supported by Replicas.run")
            else:
                out[ph] = val
        return out

    def xcd_supported(self):
        g = self.graphs[0]
        eng = g.engine
        if not hasattr(eng, "mlp_unroll_multi") or os.environ.get("L2O_DISABLE_FUSED"):
            return False
        inst = g.mlp_instance(None, dry=True)
        return inst is not None and eng.mlp_unroll_multi_supported(inst["net"].spec, inst["desc"], min(8, len(self.graphs)))

    def _draw_all(self):
        """The minibatch indices of ALL replicas in one device draw (one torch generator call instead of one per replica;
        every replica still gets its own independent index sequence) -- when no replica has a host `sampler` and the engine
        draws on the device.  Returns False when the replicas must draw for themselves."""
        graphs = self.graphs
        eng = graphs[0].engine
        if not hasattr(eng, "sample_int") or any(t.hyper.get("sampler") is not None for g in graphs for t in g.terms):
            return False
        from ._graph_core import rng
        d = graphs[0]._mlp_desc(graphs[0].terms[0])
        shape = (len(graphs), self.len_unroll + 1, d.batch)
        big = self.__dict__.get("_idx_all")
        if big is None or tuple(big.shape) != shape:
            big = self._idx_all = eng.empty_int(*shape)
        for j, g in enumerate(graphs):                       # (a graph's reset() drops its index buffers)
            g._mlp_idx = {0: big[j]}
        eng.sample_int(big, d.images.shape[0], int(rng().integers(0, 2 ** 62)))
        return True

    def launch(self, feed=None):
        """ENQUEUE one committed unroll of every replica on the one-instance-per-XCD kernel (launches of up to eight) without
        synchronising the host, without a recovery snapshot and without a status check -- the caller syncs and calls
        engine.check_unroll_status() itself (bench.py's timed region).  Returns the replicas' loss buffers (device, [T + 1])."""
        graphs = self.graphs
        eng = graphs[0].engine
        drew = self._draw_all()
        insts = [g.mlp_instance(self._feed(g, feed), draw=not drew) for g in graphs]
        if any(i is None for i in insts):
            raise _abi.L2OUnsupported(_abi.L2O_ERR_UNSUPPORTED, "Replicas.launch: l2o_mlp_unroll_multi does not apply")
        net, desc = insts[0]["net"], insts[0]["desc"]
        step0 = int(feed[graphs[0].step]) if graphs[0].rnnprop else 1
        wpack = net.wpack(eng)
        for k in range(0, len(insts), 8):
            eng.mlp_unroll_multi(net.spec, wpack, desc, insts[k:k + 8], self.len_unroll, step0)
        self.last_form = "xcd"
        for g in graphs:
            g.last_path = "mlp_xcd"
        return [i["fx"] for i in insts]

    def run(self, feed=None, form="auto"):
        """One committed unroll of every replica from its current variables (== N x sess.run([fx, update])).
        form: "xcd" (launches of up to eight instances, one per XCD), "chip" (one instance after the other on the whole
        chip) or "auto" (xcd for two or more replicas where the kernel applies).  Returns the N final losses (host)."""
        if form not in ("auto", "xcd", "chip"):
            raise ValueError("form must be auto, xcd or chip")
        graphs = self.graphs
        eng = graphs[0].engine
        use_xcd = form == "xcd" or (form == "auto" and len(graphs) > 1 and self.xcd_supported())
        if use_xcd and not self.xcd_supported():
            raise _abi.L2OUnsupported(_abi.L2O_ERR_UNSUPPORTED, "Replicas.run(form='xcd'): l2o_mlp_unroll_multi does not apply to "
                                      "this optimizee / network / device")
        T = self.len_unroll
        if not use_xcd:
            self.last_form = "chip"
            outs = [g.execute(self._feed(g, feed), True) for g in graphs]
            self.fx_arrays = [o["fx_array"] for o in outs]
            return np.array([o["fx"] for o in outs], np.float32)
        self.last_form = "xcd"
        step0 = int(feed[graphs[0].step]) if graphs[0].rnnprop else 1
        recover = not os.environ.get("L2O_NO_RECOVERY")
        insts = []
        drew = self._draw_all()
        for g in graphs:
            inst = g.mlp_instance(self._feed(g, feed), draw=not drew)
            if inst is None or (insts and (inst["desc"] is not insts[0]["desc"] or inst["net"] is not insts[0]["net"])):
                raise ValueError("Replicas.run: the replicas must be problems.mnist instances over ONE data set, stepped by "
                                 "one (20, 20) LSTM network")
            if recover:
                g._last_launch = {"restart": None, "snapshot": False, "commit": True}
                g._snapshot(g.slots)
            insts.append(inst)
        net, desc = insts[0]["net"], insts[0]["desc"]
        wpack = net.wpack(eng)
        for k in range(0, len(insts), 8):
            eng.mlp_unroll_multi(net.spec, wpack, desc, insts[k:k + 8], T, step0)
        if hasattr(eng, "prefetch_unroll_status"):
            eng.prefetch_unroll_status()
        fx_host = [eng.to_numpy(i["fx"]) for i in insts]    # host sync
        try:
            eng.check_unroll_status()
        except _abi.L2OPartnerTimeout as err:
            if not recover:
                raise
            # a team of 32 workgroups did not assemble on its XCD (a masked / shared device): every replica's inputs come
            # back and the unrolls re-run, on the SAME minibatches, on the step-granular kernels
            warnings.warn("open_l2o_amd: %s -- re-running the replicas on the step-granular kernels" % (err,), RuntimeWarning)
            self.recoveries += 1
            fx_host = []
            for g in graphs:
                snap = g._snap
                for t, b in zip(snap["live"], snap["bak"]):
                    t.copy_(b)
                g._reuse_minibatches = True
                try:
                    with _abi.option_scope({_abi.OPT_MLP_UNROLL: 0}):
                        fx, _ = g.launch(self._feed(g, feed), True, _recovering=True)
                        fx_host.append(eng.to_numpy(fx))
                finally:
                    g._reuse_minibatches = False
            self.last_form = "steps (recovered)"
        for g in graphs:
            g.last_path = "mlp_xcd"
        self.fx_arrays = fx_host
        return np.array([f[T] for f in fx_host], np.float32)
