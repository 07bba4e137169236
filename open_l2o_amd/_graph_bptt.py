"""UnrollGraph's back-propagation through time (split out of meta.py in round 4): the meta-gradient of
MetaOptimizer.meta_minimize (DM/meta.py:398-414) -- recorded history -> l2o_cwlstm_bwd_* -> l2o_cwlstm_wgrad / l2o_atb;
the less common branches (generic `layers`, Linear-only net, second derivatives) through the ABI v11 vector passes."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _abi, networks
from ._graph_core import PackedState, _DevGrad, _LazyHost, _term_vars, _world, _all_reduce, rng  # noqa: F401


class BpttMixin(object):
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
            if k not in acc:
                acc[k] = val
                return
            # acc += val (l2o_lincomb takes raw pointers: dense operands only).  The first `val` of a key is a column
            # block of a [KA, KB] contraction result -- a STRIDED view -- so the running sum moves into a
            # buffer of its own the first time something is added to it, and every later block is densified.
            a = acc[k]
            if not a.is_contiguous():
                a = acc[k] = a.contiguous()
            v = val.reshape(a.shape)
            eng.lincomb(a, a, 1.0, v if v.is_contiguous() else v.contiguous(), 1.0)

        second = any(pn.get("second") for pn in panels)

        def hess_update(pn, t, N):
            """second_derivatives: lam_t = g_t + lam_{t+1} + (d g_t / d x_t) u_t with u_t = dL/dg_t just emitted.
            g_t was recorded with its term's weight folded in (_run_steps), so the Hessian-vector product of the
            UNWEIGHTED optimizee carries the same factor."""
            hv = pn.setdefault("hv", eng.empty(N))
            eng.problem_hvp(pn["desc"], pn["xs"][t], pn["dg"].view(pn["B"], pn["D"]), hv.view(pn["B"], pn["D"]))
            w = float(pn.get("weight", 1.0))
            eng.lincomb(pn["lam"], pn["gs"][t].reshape(N), 1.0, pn["lam"], 1.0, hv, w)       # lam <- g_t + lam + w H u

        def rnnprop_input_adjoint(pn, t, N, Bt, k):
            """second_derivatives for RNNProp (DM/meta_rnnprop_train.py:380-388 without the stop_gradient): the network
            inputs m~ = m^/(sqrt(v^) + 1e-8), g~ = g/(sqrt(v^) + 1e-8) depend on g_t directly AND through the moment
            recurrences m_t = b1 m_{t-1} + (1 - b1) g_t, v_t = b2 v_{t-1} + (1 - b2) g_t^2 that later steps read.  From
            the step kernel's du (adjoint of the input projection's pre-activations) this forms u_t = dL/dg_t and the
            adjoints carried to step t - 1 (l2o_rnnprop_input_adjoint, csrc/l2o_vecops.h)."""
            eng.rnnprop_input_adjoint(Bt, 8 * H + 1, H, wdev["w_fc"], pn["gs"][t].reshape(N), pn["ms"][t].reshape(N),
                                      pn["vs"][t].reshape(N), b1 ** k, b2 ** k, b1, b2, pn["dm"], pn["dv"], pn["dg"])

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
                    buf = eng.empty(max(T, 1), N)            # ONE launch (l2o_suffix_sums) instead of T adds
                    eng.suffix_sums([pn["gs"][t].reshape(N) for t in range(T)], pn["g_final"].reshape(N), buf)
                    pn["dxs"] = [buf[t] for t in range(T)]

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
                        res = eng.empty(ka, kb)              # (blocks copied into place: buffer plumbing)
                        for r0 in range(0, ka, 96):
                            Ab = A[:, r0:r0 + 96].contiguous()
                            for c0 in range(0, kb, 176):
                                blk = eng.atb(Ab, Bmat[:, c0:c0 + 176].contiguous())
                                res[r0:r0 + blk.shape[0], c0:c0 + blk.shape[1]].copy_(blk)
                        return res
                    for l in range(nl):
                        add("lstm_%d" % (l + 1), "w_gates", atb_blocks(io["act"][l], io["dz"][l]))
                        add("lstm_%d" % (l + 1), "b_gates", eng.colsum(io["dz"][l]))
                    dd = io["dd"].view(N, 1)
                    add("linear", "w", eng.atb(io["h_last"], dd))
                    add("linear", "b", eng.colsum(dd))
                    if fc:
                        add("input_projection", "w", eng.atb(io["feats"], io["du"]))
                        add("input_projection", "b", eng.colsum(io["du"]))
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
                    add("linear", "b", eng.colsum(dd))
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
                 and hasattr(eng, "bwd_unroll")
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
            # (round 5) the A rows without their duplicated h1(t-1) / h2(t-1) columns: 17 % fewer bytes out of the BPTT kernel
            # and into the contraction (l2o_cwlstm_bwd_unroll_compact / l2o_cwlstm_wgrad_compact; bf16 contraction only)
            compact = fused and getattr(eng, "bwd_unroll_compact", False) and not _abi.get_option(_abi.OPT_EXACT_GATES)
            if fused:                                       # all T steps in one launch, the carries in registers;
                A = eng.empty(T + 1, R, KA - 2 * H) if compact else eng.empty(T, R, KA)
                Bm = eng.empty(T, R, KB)                    # the kernel writes every row (padding rows as zeros)
                tkey = ("bwd_table", id(net), T)
                table = None if cache is None else cache.get(tkey)
                if table is None:
                    table = eng.bwd_table(grp, T) if hasattr(eng, "bwd_table") else None
                    if cache is not None:
                        cache[tkey] = table
                eng.bwd_unroll(spec, wdev, grp, T, step0, A, Bm, table=table, **({"compact": True} if compact else {}))
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
            Gm = eng.wgrad_compact(spec, A, Bm) if compact else eng.wgrad(spec, A.view(T * R, KA), Bm.view(T * R, KB))
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
            for acc in out.values():
                keys = sorted(acc)
                flat = torch.cat([acc[k].reshape(-1) for k in keys])
                _all_reduce(flat)
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

