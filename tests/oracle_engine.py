"""An engine with the interface of open_l2o_amd._engine.HipEngine whose arithmetic is the
NumPy oracle, on CPU torch tensors.  TEST INFRASTRUCTURE ONLY: it lets the CPU suite
exercise the host logic of open_l2o_amd.meta (session semantics, net assignments,
checkpoints, batch sharding + all-reduce under gloo) without a GPU.  It lives under
tests/ -- the package has no CPU compute path."""
import numpy as np
import torch

import oracle as O
from open_l2o_amd import _abi
from open_l2o_amd._engine import pack_weights_host


def _cfg_of(spec):
    if spec.kind == _abi.NET_RNNPROP:
        return O.NetConfig("rnnprop", tuple(spec.layers), "fc", {"dim": 20}, spec.scale, spec.tanh_output)
    if spec.preprocess == _abi.PRE_LOGSIGN:
        return O.NetConfig("cw", tuple(spec.layers), "LogAndSign", {"k": spec.logsign_k}, spec.scale,
                           spec.tanh_output)
    return O.NetConfig("cw", tuple(spec.layers), "identity", None, spec.scale, spec.tanh_output)


def pack_state(h1, c1, h2, c2, B, D):
    """The documented packed layout, vectorised."""
    tpp = (D + 15) // 16
    ref = np.zeros((4, B, tpp * 16, 20), np.float32)
    for a, arr in enumerate((h1, c1, h2, c2)):
        ref[a, :, :D] = np.asarray(arr, np.float32).reshape(B, D, 20)
    p = ref.reshape(4, B, tpp, 16, 5, 4)                 # a, b, tw, c, t, q
    p = p.transpose(1, 2, 0, 4, 5, 3)                    # b, tw, a, t, q, c
    p = p.reshape(B, tpp, 20, 64)                        # e = a*5+t ; lane = q*16+c
    p = p.reshape(B, tpp, 5, 4, 64).transpose(0, 1, 2, 4, 3)   # b, tw, j, lane, w
    return np.ascontiguousarray(p).reshape(-1)


def unpack_state(st, B, D):
    tpp = (D + 15) // 16
    p = np.asarray(st, np.float32).reshape(B, tpp, 5, 64, 4).transpose(0, 1, 2, 4, 3)
    p = p.reshape(B, tpp, 4, 5, 4, 16)                   # b, tw, a, t, q, c
    p = p.transpose(2, 0, 1, 5, 3, 4).reshape(4, B, tpp * 16, 20)
    return [np.ascontiguousarray(p[a, :, :D]).reshape(B * D, 20) for a in range(4)]


class OracleEngine(object):
    name = "oracle"

    def __init__(self):
        self.device = torch.device("cpu")
        self.lib = _abi.lib()
        self.calls = []
        # the product engine's status-word protocol, emulated: `fused` unrolls of this engine "exchange" (so the host's
        # snapshot / recovery / status all-reduce logic runs in the CPU suite); an injected fault makes the next such
        # unroll leave garbage and raise the sticky status, exactly once
        self._status = torch.zeros(1, dtype=torch.int32)
        self._fault = False
        self._exchanged = False

    def inject_unroll_fault(self, on=True):
        self._fault = bool(on)

    def last_unroll_exchanges(self):
        return self._exchanged

    def unroll_status_tensor(self):
        return self._status

    def check_unroll_status(self):
        if int(self._status[0]):
            self._status.zero_()
            self._fault = False
            raise _abi.L2OPartnerTimeout(_abi.L2O_ERR_TIMEOUT, "oracle engine: injected partner timeout")

    def atb(self, A, B):
        """A^T B (the HIP engine's l2o_atb), CPU torch."""
        return A.t().contiguous() @ B

    def wgrad(self, spec, A, B):
        """The HIP engine's l2o_cwlstm_wgrad: the callers read the weight-gradient blocks of A^T B only."""
        return self.atb(A, B)

    # the HIP engine's ABI v11 vector passes (csrc/l2o_vecops.h), CPU torch restatements of the same formulas
    @staticmethod
    def _dense(*ts):
        """HipEngine passes raw pointers (_engine._ptr asserts is_contiguous()): the CPU stand-ins hold their operands
        to the same contract, so a strided view that would trip the GPU path trips the CPU suite too."""
        for t in ts:
            assert t is None or (t.is_contiguous() and t.dtype == torch.float32), \
                "vector passes take dense fp32 operands (shape %r, stride %r)" % (tuple(t.shape), t.stride())

    def suffix_sums(self, gs, g_final, out):
        self._dense(g_final, out, *gs)
        acc = g_final.reshape(-1).clone()
        for t in reversed(range(len(gs))):
            out[t] = acc
            acc = acc + gs[t].reshape(-1)
        return out

    def colsum(self, A, out=None, accumulate=False):
        self._dense(A, out)
        r = A.sum(dim=-2)
        if out is None:
            return r
        out.copy_(out + r if accumulate else r)
        return out

    def lincomb(self, out, a, ca=1.0, b=None, cb=0.0, c=None, cc=0.0):
        self._dense(out, a, b, c)
        r = ca * a.reshape(-1)
        if b is not None:
            r = r + cb * b.reshape(-1)
        if c is not None:
            r = r + cc * c.reshape(-1)
        out.reshape(-1).copy_(r)
        return out

    def rnnprop_input_adjoint(self, Bm, du_col, H, w_fc, g, m, v, pow1, pow2, beta1, beta2, dm, dv, dg):
        self._dense(w_fc, g, m, v, dm, dv, dg)
        f = np.float32
        du = Bm[:g.numel(), du_col:du_col + H]
        wfc = w_fc.view(2, H)
        a0, a1 = (du * wfc[0]).sum(1), (du * wfc[1]).sum(1)
        g, m, v = g.reshape(-1), m.reshape(-1), v.reshape(-1)
        om1, om2 = float(f(1.0 - pow1)), float(f(1.0 - pow2))
        m_hat, sq = m / om1, torch.sqrt(v / om2)
        den = sq + 1e-8
        d_den = -(a0 * m_hat + a1 * g) / (den * den)
        d_vhat = torch.where(sq > 0, d_den * 0.5 / sq.clamp_min(1e-30), torch.zeros_like(sq))
        dmn = a0 / den / om1 + dm
        dvn = d_vhat / om2 + dv
        dg.copy_(a1 / den + dmn * float(f(1.0) - f(beta1)) + dvn * (2.0 * float(f(1.0) - f(beta2))) * g)
        dm.copy_(dmn * float(f(beta1)))
        dv.copy_(dvn * float(f(beta2)))

    def tensor(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).copy())

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32)

    def empty(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32)

    def to_numpy(self, t):
        return t.detach().numpy().copy()

    def pack_weights(self, spec, params):
        # run the real host packer (validates shapes) but keep the .l2l dict for the oracle
        t = self.tensor(pack_weights_host(self.lib, spec, params))
        t._l2l = {k: {v: np.array(a, np.float32) for v, a in d.items()} for k, d in params.items()}
        return t

    # the meta-step "on the device" (HipEngine.adam_step / pack_weights_device): same contract on CPU
    # tensors, so that the CPU suite runs the host logic of that path (gradient layout, lazy .l2l refresh)
    _W_NAMES = {"w_gates1": ("lstm_1", "w_gates"), "b_gates1": ("lstm_1", "b_gates"),
                "w_gates2": ("lstm_2", "w_gates"), "b_gates2": ("lstm_2", "b_gates"),
                "w_lin": ("linear", "w"), "b_lin": ("linear", "b"),
                "w_fc": ("input_projection", "w"), "b_fc": ("input_projection", "b")}

    def adam_step(self, w, m, v, g, lr_t, beta1, beta2, epsilon, guarded=False):
        if guarded and int(self._status[0]):                # l2o_adam_step_guarded: the failed unroll's update is skipped
            self.calls.append("adam_step_skipped")
            return
        self.calls.append("adam_step")
        f = np.float32
        wn, mn, vn, gn = w.numpy(), m.numpy(), v.numpy(), g.numpy()      # views: in place
        mn[:] = f(beta1) * mn + f(1.0 - beta1) * gn
        vn[:] = f(beta2) * vn + f(1.0 - beta2) * gn * gn
        wn[:] = wn - f(lr_t) * mn / (np.sqrt(vn) + f(epsilon))

    def pack_weights_device(self, spec, weights, out):
        self.calls.append("pack_weights_device")
        params = {}
        for k, (mod, var) in self._W_NAMES.items():
            if weights.get(k) is not None:
                params.setdefault(mod, {})[var] = weights[k].numpy().copy()
        out.copy_(torch.from_numpy(pack_weights_host(self.lib, spec, params)))
        out._l2l = params

    def state_floats(self, B, D):
        return int(self.lib.l2o_state_floats(B, D))

    def state_alloc(self, B, D):
        return self.zeros(self.state_floats(B, D))

    def state_pack(self, h1, c1, h2, c2, B, D):
        return self.tensor(pack_state(h1.numpy(), c1.numpy(), h2.numpy(), c2.numpy(), B, D))

    def state_unpack(self, st, B, D, H=20):
        return [self.tensor(a) for a in unpack_state(st.numpy(), B, D)]

    # ------------------------------------------------------------------ compute
    def _oracle_problem(self, p):
        n = lambda t: None if t is None else t.numpy()
        if getattr(p, "w_shared", False):                   # one matrix for every problem: materialise the batch
            import dataclasses
            Wfull = np.broadcast_to(n(p.W).reshape(1, p.M, p.D), (p.B_local, p.M, p.D)).copy()
            p = dataclasses.replace(p, W=torch.from_numpy(Wfull), w_shared=False)
        if p.kind == _abi.PROB_SIMPLE:
            return O.SimpleMulti(p.D), (p.D,)
        if p.kind == _abi.PROB_QUADRATIC:
            return O.Quadratic(n(p.W).reshape(p.B_local, p.M, p.D), n(p.y).reshape(p.B_local, p.M),
                               batch_global=p.B_global), (p.B_local, p.D)
        if p.kind == _abi.PROB_LASSO:
            return O.Lasso(n(p.W).reshape(p.B_local, p.M, p.D), n(p.y).reshape(p.B_local, p.M, 1), p.l1,
                           batch_global=p.B_global), (p.B_local, p.D)
        if p.kind == _abi.PROB_RASTRIGIN:
            return O.Rastrigin(n(p.W).reshape(p.B_local, p.D, p.D), n(p.y).reshape(p.B_local, p.D, 1),
                               n(p.C).reshape(p.B_local, p.D, 1), p.alpha,
                               batch_global=p.B_global), (p.B_local, p.D, 1)
        if p.kind == _abi.PROB_SQUARE_COS:
            # the engine interface carries the column sums of wcos: any wcos with those column
            # sums gives the same loss and gradient -> put them in its first row
            wcos = np.zeros((p.B_local, p.D, p.D), np.float32)
            wcos[:, 0, :] = n(p.C).reshape(p.B_local, p.D)
            return O.SquareCos(n(p.W).reshape(p.B_local, p.D, p.D), n(p.y).reshape(p.B_local, p.D), wcos,
                               batch_global=p.B_global), (p.B_local, p.D)
        raise _abi.L2OUnsupported(-2, "kind %d" % p.kind)

    def problem_fg(self, p, x, f_part, g):
        self.calls.append("problem_fg")
        prob, shape = self._oracle_problem(p)
        xs = x.numpy().reshape(p.B_local, p.D)
        s = None if p.x_scale is None else p.x_scale.numpy().reshape(p.B_local, p.D)
        xin = (xs if s is None else xs * s).reshape(shape)
        if p.kind == _abi.PROB_SIMPLE:
            f_part.copy_(torch.from_numpy(np.array([prob.f(xin)], np.float32)))
            gr = prob.grad(xin)
        else:
            f_part.copy_(torch.from_numpy(prob.f_per_problem(xin).astype(np.float32)))
            gr = prob.grad(xin)
        if g is not None:
            gr = gr.reshape(p.B_local, p.D)
            if s is not None:
                gr = gr * s
            g.copy_(torch.from_numpy(gr.astype(np.float32)).view_as(g))

    def problem_hvp(self, p, x, u, out):
        """out = (d g / d x) u with g what problem_fg returns (l2o_problem_hvp, include/l2o_abi.h): the closed forms of
        the analytic optimizees' Hessians (DM/problems.py:98-99, 128-131, 206-211 differentiated twice), 1/B_global and
        the x_scale chain rule included; the l1 term of lasso has no curvature."""
        self.calls.append("problem_hvp")
        self._dense(x, u, out)
        f = np.float32
        B, D = p.B_local, p.D
        xs, us = x.numpy().reshape(B, D), u.numpy().reshape(B, D)
        s = np.ones((B, D), f) if p.x_scale is None else p.x_scale.numpy().reshape(B, D)
        su = s * us
        if p.kind == _abi.PROB_SIMPLE:
            r = f(2.0) * s * su
        else:
            W = p.W.numpy().reshape((1 if getattr(p, "w_shared", False) else B), p.M, D)
            r = np.einsum("bmd,bm->bd", W, np.einsum("bmd,bd->bm", W, su))
            if p.kind in (_abi.PROB_QUADRATIC, _abi.PROB_SQUARE_COS):      # ||Wx - y||^2 without the 1/2
                r = f(2.0) * r
            if p.kind in (_abi.PROB_RASTRIGIN, _abi.PROB_SQUARE_COS):
                two_pi = f(2.0 * np.pi)
                r = r + two_pi * two_pi * f(p.alpha) * p.C.numpy().reshape(B, D) * np.cos(two_pi * xs * s) * su
            r = s * r / f(p.B_global)
        out.copy_(torch.from_numpy(np.ascontiguousarray(r, f)).view_as(out))

    def int_tensor(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32).copy())

    def mlp_fg(self, d, indices, w1, b1, w2, b2, loss, grads):
        self.calls.append("mlp_fg")
        prob = O.MnistMLP(d.images.numpy(), d.labels.numpy(), "sigmoid" if d.activation == 0 else "relu")
        variables = [w1.numpy().reshape(d.n_in, d.n_hidden), b1.numpy().reshape(-1),
                     w2.numpy().reshape(d.n_hidden, d.n_out), b2.numpy().reshape(-1)]
        f, g = prob.fg(variables, indices.numpy(), want_grad=grads is not None)
        loss.copy_(torch.from_numpy(np.array([f], np.float32)))
        if grads is not None:
            for t, a in zip(grads, g):
                t.copy_(torch.from_numpy(np.ascontiguousarray(a, np.float32)).view_as(t))

    def mlp_deep_fg(self, d, indices, ws, loss, grads):
        self.calls.append("mlp_deep_fg")
        prob = O.MnistMLP(d.images.numpy(), d.labels.numpy(), "sigmoid" if d.activation == 0 else "relu")
        widths = [d.n_in] + list(d.hidden) + [d.n_out]
        variables = []
        for l in range(len(widths) - 1):
            variables += [ws[2 * l].numpy().reshape(widths[l], widths[l + 1]), ws[2 * l + 1].numpy().reshape(-1)]
        f, g = prob.fg_deep(variables, indices.numpy(), want_grad=grads is not None)
        loss.copy_(torch.from_numpy(np.array([f], np.float32)))
        if grads is not None:
            for t, a in zip(grads, g):
                t.copy_(torch.from_numpy(np.ascontiguousarray(a, np.float32)).view_as(t))

    def lstm_step_multi(self, spec, wpack, segs, pow1, pow2):
        for seg in segs:
            g, m, v, st, x, B, D = seg[:7]
            if len(seg) > 7:                               # out of place: inputs stay, results into *_out
                st_out, m_out, v_out = seg[7:10]
                if st is not None:
                    st_out.copy_(st)
                if m is not None:
                    m_out.copy_(m); v_out.copy_(v)
                st, m, v = (st_out if st is not None else None), (m_out if m is not None else None), \
                           (v_out if v is not None else None)
            self.lstm_step(spec, wpack, g, m, v, pow1, pow2, st, x, B, D)

    def lstm_step(self, spec, wpack, g, m, v, pow1, pow2, st, x, B, D):
        self.calls.append("lstm_step")
        cfg = _cfg_of(spec)
        params = wpack._l2l
        gn = g.numpy().reshape(B, D)
        if len(cfg.layers):
            h1, c1, h2, c2 = unpack_state(st.numpy(), B, D)
            state = ((h1, c1), (h2, c2))
        else:
            state = ()
        dt = np.float32
        if cfg.kind == "rnnprop":
            mn = dt(spec.beta1) * m.numpy() + dt(1.0 - spec.beta1) * gn
            vn = dt(spec.beta2) * v.numpy() + dt(1.0 - spec.beta2) * gn * gn
            mh = mn / (dt(1) - dt(pow1))
            vh = vn / (dt(1) - dt(pow2))
            inputs = (mh / (np.sqrt(vh) + dt(1e-8)), gn / (np.sqrt(vh) + dt(1e-8)))
            m.copy_(torch.from_numpy(mn))
            v.copy_(torch.from_numpy(vn))
        else:
            inputs = gn
        delta, nstate = O.net_apply(cfg, params, inputs, state)
        x.add_(torch.from_numpy(delta.astype(np.float32)).view_as(x))
        if len(cfg.layers):
            st.copy_(torch.from_numpy(pack_state(nstate[0][0], nstate[0][1], nstate[1][0], nstate[1][1], B, D)))

    def bwd_step(self, spec, weights, io, pow1, pow2, B, D):
        self.calls.append("bwd_step")
        cfg = _cfg_of(spec)
        N = B * D
        names = {"w_gates1": ("lstm_1", "w_gates"), "b_gates1": ("lstm_1", "b_gates"),
                 "w_gates2": ("lstm_2", "w_gates"), "b_gates2": ("lstm_2", "b_gates"),
                 "w_lin": ("linear", "w"), "b_lin": ("linear", "b"),
                 "w_fc": ("input_projection", "w"), "b_fc": ("input_projection", "b")}
        params = {}
        for k, (m, n) in names.items():
            if weights.get(k) is not None:
                params.setdefault(m, {})[n] = weights[k].numpy()
        g = io["g"].numpy().reshape(N)
        dt = np.float32
        if cfg.kind == "rnnprop":
            mh = io["m"].numpy().reshape(N) / (dt(1) - dt(pow1))
            vh = io["v"].numpy().reshape(N) / (dt(1) - dt(pow2))
            inputs = (mh / (np.sqrt(vh) + dt(1e-8)), g / (np.sqrt(vh) + dt(1e-8)))
        else:
            inputs = g
        if len(cfg.layers):
            h1, c1, h2, c2 = unpack_state(io["st_prev"].numpy(), B, D)
            state = ((h1, c1), (h2, c2))
            cin = io["carry_in"].numpy().reshape(4, N, 20)
            carry = tuple(cin[i] for i in range(4))
        else:
            state, carry = (), None
        cout, rows = O.net_bwd_step(cfg, params, inputs, state, io["dx_next"].numpy().reshape(N), carry)
        if cout is not None:
            io["carry_out"].copy_(torch.from_numpy(np.stack(cout).astype(np.float32)).view_as(io["carry_out"]))
        da = rows.pop("da", None)
        if io.get("dg") is not None:                               # u_t = dL/dg_t (second_derivatives, DM nets)
            io["dg"].copy_(torch.from_numpy(np.ascontiguousarray(O.preprocess_bwd(cfg, g, da), np.float32)).view_as(io["dg"]))
        for k, v in rows.items():
            t = io[k]
            a = np.ascontiguousarray(v, np.float32)
            if k == "act1" and not len(cfg.layers):                 # [N,P] -> padded [N,2]
                pad = np.zeros((N, 2), np.float32)
                pad[:, :a.shape[1]] = a
                a = pad
            t.copy_(torch.from_numpy(a).view_as(t))

    def bwd_multi(self, spec, weights, segs, carry_in, carry_out, A, Bm, pow1, pow2):
        """Same contract as HipEngine.bwd_multi, through bwd_step on row-block views."""
        fc = spec.preprocess == _abi.PRE_FC_ELU
        P = 20 if fc else (2 if spec.preprocess == _abi.PRE_LOGSIGN else 1)
        H, K1 = 20, P + 20
        row = 0
        for sg in segs:
            B, D = sg["B"], sg["D"]
            N = B * D
            At, Bt = A[row:row + N], Bm[row:row + N]
            cin = carry_in[:, row:row + N].contiguous()
            cout = torch.empty_like(cin)
            io = dict(g=sg["g"], m=sg.get("m"), v=sg.get("v"), st_prev=sg["st_prev"], dx_next=sg["dx_next"],
                      carry_in=cin, carry_out=cout, act1=At[:, 0:K1], act2=At[:, K1:K1 + 2 * H],
                      h2=At[:, K1 + 2 * H:K1 + 3 * H], dz1=Bt[:, 0:4 * H], dz2=Bt[:, 4 * H:8 * H],
                      dd=Bt[:, 8 * H:8 * H + 1])
            if fc:
                io.update(feats=At[:, K1 + 3 * H:K1 + 3 * H + 2], du=Bt[:, 8 * H + 1:8 * H + 1 + H])
            self.bwd_step(spec, weights, io, pow1, pow2, B, D)
            carry_out[:, row:row + N] = cout
            row += (N + 15) // 16 * 16

    bwd_unroll_compact = True

    @staticmethod
    def _compact_geom(spec):
        fc = spec.preprocess == _abi.PRE_FC_ELU
        P = 20 if fc else (2 if spec.preprocess == _abi.PRE_LOGSIGN else 1)
        return P, P + 20 + 60 + (2 if fc else 0) + 1

    def wgrad_compact(self, spec, Ac, Bm):
        """HipEngine.wgrad_compact: the [KA, KB] product over the VIRTUAL columns [in | h1(t-1) | h1(t) | h2(t-1) | h2(t) |
        feats | 1] of the compact rows (block t + 1 = step t, block 0 = the state before step 0)."""
        T, R, KB = Bm.shape
        P, KA = self._compact_geom(spec)
        assert tuple(Ac.shape) == (T + 1, R, KA - 40)
        cur, prev = Ac[1:], Ac[:-1]
        A = torch.cat([cur[..., :P], prev[..., P:P + 20], cur[..., P:P + 20], prev[..., P + 20:P + 40], cur[..., P + 20:P + 40],
                       cur[..., P + 40:]], dim=-1)
        return self.atb(A.reshape(T * R, KA), Bm.reshape(T * R, KB))

    def bwd_unroll(self, spec, weights, panels, T, step0, A, Bm, carry_in=None, carry_out=None, table=None, compact=False):
        """Same contract as HipEngine.bwd_unroll, step by step through bwd_multi."""
        if compact:                                        # the full rows, then packed: block t + 1 <- [in | h1(t) | h2(t) | rest]
            P, KA = self._compact_geom(spec)
            full = torch.zeros(T, A.shape[1], KA)
            self.bwd_unroll(spec, weights, panels, T, step0, full, Bm, carry_in, carry_out, table)
            A.zero_()
            A[1:] = torch.cat([full[..., :P], full[..., P + 20:P + 40], full[..., P + 60:P + 80], full[..., P + 80:]], dim=-1)
            A[0, :, P:P + 20] = full[0][:, P:P + 20]        # h1, h2 BEFORE step 0
            A[0, :, P + 20:P + 40] = full[0][:, P + 40:P + 60]
            return
        R = A.shape[1]
        b1, b2 = float(np.float32(spec.beta1)), float(np.float32(spec.beta2))
        cin = torch.zeros(4, R, 20) if carry_in is None else carry_in.clone()
        cout = torch.zeros(4, R, 20)
        acc = [None if pn.get("g_final") is None else pn["g_final"].reshape(-1).clone() for pn in panels]
        A.zero_(); Bm.zero_()                              # the kernel writes every row: padding rows as zeros,
        row = 0
        for pn in panels:                                  # the ones column on the rows that exist
            n = pn["B"] * pn["D"]
            A[:, row:row + n, A.shape[2] - 1] = 1.0
            row += (n + 15) // 16 * 16
        for t in reversed(range(T)):
            segs = []
            for i, pn in enumerate(panels):
                dx = pn["dxs"][t] if pn.get("dxs") else acc[i].clone()
                segs.append(dict(g=pn["gs"][t], m=pn["ms"][t] if pn.get("ms") else None,
                                 v=pn["vs"][t] if pn.get("vs") else None, st_prev=pn["sts"][t], dx_next=dx,
                                 B=pn["B"], D=pn["D"]))
                if acc[i] is not None:
                    acc[i] = acc[i] + pn["gs"][t].reshape(-1)
            k = step0 + t
            self.bwd_multi(spec, weights, segs, cin, cout, A[t], Bm[t], b1 ** k, b2 ** k)
            cin, cout = cout, cin
        if carry_out is not None:
            carry_out.copy_(cin)

    def unroll_supported(self, spec, p, record=False):
        cc = spec.to_c()
        import ctypes as C
        cp = _abi.Problem()
        cp.kind, cp.B_local, cp.B_global, cp.D, cp.M = p.kind, p.B_local, p.B_global, p.D, p.M
        fn = self.lib.l2o_unroll_record_supported if record else self.lib.l2o_unroll_supported
        return bool(fn(C.byref(cc), C.byref(cp)))

    def unroll(self, spec, wpack, p, x, st, m, v, T, step0, fx_part, hist=None, fx=None, x0=None, zero_state=False):
        if x0 is not None:
            x.copy_(x0)
        if zero_state:
            st.zero_()
            if m is not None:
                m.zero_(); v.zero_()
        self._exchanged = bool(_abi.get_option(_abi.OPT_PAIR))      # (OPT_PAIR = 0: the exchange-free kernels)
        if self._fault and self._exchanged:                          # a partner timeout: outputs are garbage
            self.calls.append("unroll_timeout")
            x.fill_(float("nan")); st.fill_(float("nan")); fx_part.fill_(float("nan"))
            if fx is not None:
                fx.fill_(float("nan"))
            if hist is not None:
                for t in hist.values():
                    if t is not None:
                        t.fill_(float("nan"))
            self._status.fill_(1)
            return
        self._unroll(spec, wpack, p, x, st, m, v, T, step0, fx_part, hist)
        if fx is not None:                                         # l2o_unroll_reduce
            self.reduce_fx(fx_part, T + 1, p.B_local, p.B_global, fx)

    def _unroll(self, spec, wpack, p, x, st, m, v, T, step0, fx_part, hist=None):
        self.calls.append("unroll")
        cfg = _cfg_of(spec)
        prob, shape = self._oracle_problem(p)
        B, D = p.B_local, p.D
        h1, c1, h2, c2 = unpack_state(st.numpy(), B, D)
        s = None if p.x_scale is None else p.x_scale.numpy().reshape(shape)
        xcur = x.numpy().reshape(shape).copy()
        state = ((h1, c1), (h2, c2))
        mm = None if m is None else m.numpy().reshape(shape).copy()
        vv = None if v is None else v.numpy().reshape(shape).copy()
        fparts = np.zeros((T + 1, B), np.float32)
        # step-by-step so that per-problem losses are available
        for t in range(T + 1):
            xin = xcur if s is None else xcur * s
            fparts[t] = prob.f_per_problem(xin)
            if hist is not None:                                   # what l2o_unroll_record stores
                gt = prob.grad(xin) if s is None else prob.grad(xin) * s
                if t == T:
                    hist["g_final"].copy_(torch.from_numpy(gt.reshape(-1)).view_as(hist["g_final"]))
                else:
                    hist["g"][t].copy_(torch.from_numpy(gt.reshape(-1)).view_as(hist["g"][t]))
                    hist["st"][t].copy_(torch.from_numpy(pack_state(state[0][0], state[0][1], state[1][0],
                                                                    state[1][1], B, D)).view_as(hist["st"][t]))
            if t == T:
                break
            res = O.unroll(prob, cfg, wpack._l2l, xcur, state, 1, x_scale=s, m0=mm, v0=vv,
                           step0=step0 + t, beta1=spec.beta1, beta2=spec.beta2)
            xcur, state, mm, vv = res.x, res.state, res.m, res.v
            if hist is not None and mm is not None and hist.get("m") is not None:
                hist["m"][t].copy_(torch.from_numpy(mm.reshape(-1)).view_as(hist["m"][t]))
                hist["v"][t].copy_(torch.from_numpy(vv.reshape(-1)).view_as(hist["v"][t]))
        x.copy_(torch.from_numpy(xcur.reshape(B, D)))
        st.copy_(torch.from_numpy(pack_state(state[0][0], state[0][1], state[1][0], state[1][1], B, D)))
        if mm is not None and m is not None:
            m.copy_(torch.from_numpy(mm.reshape(B, D)))
            v.copy_(torch.from_numpy(vv.reshape(B, D)))
        fx_part.copy_(torch.from_numpy(fparts.reshape(-1)))

    def reduce_fx(self, fx_part, T1, B_local, B_global, fx):
        fp = fx_part.numpy().reshape(T1, B_local)
        fx.copy_(torch.from_numpy((fp.sum(axis=1, dtype=np.float32) / np.float32(B_global)).astype(np.float32)))

    def synchronize(self):
        pass
