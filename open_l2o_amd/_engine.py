"""Execution engine: torch tensors for device memory + streams, the HIP C-ABI
for every bit of arithmetic on the hot path.

The host layer (``meta.py``) talks to an *engine* object.  The product engine is
:class:`HipEngine`; it needs a GPU and ``libl2o_hip.so`` and raises otherwise.
Tests may inject a different engine (e.g. an oracle-backed one that lives under
``tests/``) to exercise the host logic and the multi-process sharding on CPU --
the package itself contains no CPU compute path.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import Optional

import numpy as np
import torch

from . import _abi


@dataclasses.dataclass
class NetSpec:
    """The ``net_options`` of one optimizer network (DM/networks.py:157-159) in
    the form the kernels consume."""
    kind: int                 # _abi.NET_*
    preprocess: int           # _abi.PRE_*
    layers: tuple
    scale: float = 1.0
    tanh_output: bool = False
    logsign_k: float = 0.0
    beta1: float = 0.95
    beta2: float = 0.95

    @property
    def generic(self):
        """An LSTM stack the matrix-core kernels do not implement (they cover the harness' (20, 20)): stepped by
        l2o_cwlstm_step_generic, state in the per-layer [N, H] layout (l2o_gen_state_floats)."""
        return len(self.layers) > 0 and tuple(int(h) for h in self.layers) != (20, 20)

    def to_c(self):
        c = _abi.NetCfg()
        c.kind, c.preprocess = self.kind, self.preprocess
        c.n_layers = len(self.layers)
        c.hidden = int(self.layers[0]) if self.layers else 0
        if len(self.layers) > 1 and any(int(h) != c.hidden for h in self.layers):
            c.hidden = -1
        c.tanh_output = 1 if self.tanh_output else 0
        c.scale, c.logsign_k = float(self.scale), float(self.logsign_k)
        c.beta1, c.beta2 = float(self.beta1), float(self.beta2)
        c.options = _abi.options_word()             # the caller-owned kernel switches travel with every call
        return c


@dataclasses.dataclass
class ProblemDesc:
    """Device-side view of one optimizee batch shard (struct l2o_problem)."""
    kind: int
    B_local: int
    B_global: int
    D: int
    M: int = 0
    l1: float = 0.0
    alpha: float = 0.0
    W: Optional[torch.Tensor] = None
    y: Optional[torch.Tensor] = None
    C: Optional[torch.Tensor] = None
    x_scale: Optional[torch.Tensor] = None
    w_shared: bool = False          # W is ONE [M, D] matrix for every problem (L2O_PROB_W_SHARED)


@dataclasses.dataclass
class MlpDesc:
    """Device-side view of problems.mnist (struct l2o_mlp)."""
    n_in: int
    n_hidden: int
    n_out: int
    batch: int
    activation: int           # 0 sigmoid, 1 relu
    images: torch.Tensor      # [n_data, n_in] fp32
    labels: torch.Tensor      # [n_data] int32


@dataclasses.dataclass
class MlpDeepDesc:
    """Device-side view of problems.mnist with several hidden layers (struct l2o_mlp_deep)."""
    n_in: int
    hidden: tuple
    n_out: int
    batch: int
    activation: int
    images: torch.Tensor
    labels: torch.Tensor


def _ptr(t):
    if t is None:
        return None
    assert t.dtype == torch.float32 and t.is_contiguous(), "C-ABI wants contiguous fp32"
    return C.c_void_p(t.data_ptr())


class HipEngine(object):
    """MI355X engine: every method is one call through the C ABI."""

    name = "hip"

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("open_l2o_amd.HipEngine needs a ROCm GPU (torch.cuda.is_available() is "
                               "False); there is no CPU fallback")
        self.lib = _abi.lib()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self._workspace = None
        self._last_ws = None
        self._mlp_scratch = None
        # how many one-per-CU workgroups are really co-resident on this device (CU masks below HIP's view, shared
        # partitions): measured ONCE per device, cached by the library; the two-CU unroll and l2o_mlp_unroll size their
        # launches against it (l2o_coresident_workgroups)
        with torch.cuda.device(self.device):
            scratch = torch.zeros(16, dtype=torch.int32, device=self.device)
            self.coresident_cus = int(self.lib.l2o_coresident_workgroups(C.c_void_p(scratch.data_ptr()), self._stream()))
        if self.coresident_cus <= 0:
            _abi.check(self.coresident_cus)

    # -- memory plumbing (torch) ------------------------------------------
    def tensor(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def sample(self, kind, shape, a, b, seed):
        """A random fp32 device tensor: kind "normal" (mean a, stddev b) or "uniform" ([a, b)), from a device generator
        seeded with `seed` (problem re-sampling of MetaLoss.reset without a host draw + upload)."""
        g = self.__dict__.get("_gen")
        if g is None:
            g = self._gen = torch.Generator(device=self.device)
        g.manual_seed(int(seed))
        if kind == "normal":
            t = torch.randn(shape, generator=g, device=self.device, dtype=torch.float32)
            return t.mul_(float(b)).add_(float(a))
        t = torch.rand(shape, generator=g, device=self.device, dtype=torch.float32)
        return t.mul_(float(b) - float(a)).add_(float(a))

    def sample_int(self, out, high, seed):
        """Uniform int32 draws in [0, high) into the device tensor `out`, from a device generator seeded with `seed`: the
        minibatch index rows of the neural optimizee (DM/problems.py:282-284 draws them with tf.random_uniform -- a device
        op there too) without a host draw + upload, which as a pageable H2D copy waited for the previous unroll to
        finish (host enqueue 2.1 ms per 2.3 ms config-5 unroll)."""
        g = self.__dict__.get("_gen_int")
        if g is None:
            g = self._gen_int = torch.Generator(device=self.device)
        g.manual_seed(int(seed))
        return torch.randint(0, int(high), tuple(out.shape), generator=g, device=self.device, dtype=torch.int32, out=out)

    def empty_int(self, *shape):
        return torch.empty(*shape, dtype=torch.int32, device=self.device)

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.device)

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def to_numpy(self, t):
        return t.detach().cpu().numpy()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- weights -------------------------------------------------------------
    def pack_weights(self, spec: NetSpec, params: dict, key=None):
        """.l2l dict (Sonnet layouts) -> device wpack.  key: see upload()."""
        host = pack_weights_host(self.lib, spec, params)
        return self.tensor(host) if key is None else self.upload(key, host)

    def pack_weights_device(self, spec: NetSpec, weights: dict, out):
        """Device Sonnet-layout weights (dict keyed like struct l2o_net_weights) -> `out` (device wpack),
        without leaving the device (l2o_wpack_device; bit-equal to pack_weights)."""
        cc = spec.to_c()
        w = _abi.NetWeights()
        for k, _ in _abi.NetWeights._fields_:
            setattr(w, k, None if (k == "wpack" or weights.get(k) is None) else weights[k].data_ptr())
        # a buffer this engine has packed before for the same configuration: its padding words are zero already
        # (L2O_OPT_WPACK_NO_CLEAR: no memset in front of the pack kernel on the meta-step's critical path)
        sig = (spec.preprocess, tuple(spec.layers), spec.kind)
        if getattr(out, "_l2o_packed", None) == sig:
            cc.options |= (8 | 1) << (4 * _abi.OPT_WPACK_NO_CLEAR)
        _abi.check(self.lib.l2o_wpack_device(C.byref(cc), C.byref(w), _ptr(out), self._stream()))
        try:
            out._l2o_packed = sig
        except AttributeError:
            pass

    def adam_step(self, w, m, v, g, lr_t, beta1, beta2, epsilon, guarded=False):
        """TF-1.x Adam on one flat device vector, in place (l2o_adam_step).  guarded: conditional, ON THE DEVICE, on the
        status word of the last fused unroll's workspace (l2o_adam_step_guarded) -- a partner timeout of that unroll
        leaves w, m, v untouched; the caller can then enqueue the update before it has seen the status on the host."""
        ws = self._last_ws if guarded else None
        if ws is not None:
            _abi.check(self.lib.l2o_adam_step_guarded(_ptr(w), _ptr(m), _ptr(v), _ptr(g), int(w.numel()), float(lr_t),
                                                      float(beta1), float(beta2), float(epsilon), C.c_void_p(ws.data_ptr()),
                                                      self._stream()))
        else:
            _abi.check(self.lib.l2o_adam_step(_ptr(w), _ptr(m), _ptr(v), _ptr(g), int(w.numel()), float(lr_t),
                                              float(beta1), float(beta2), float(epsilon), self._stream()))

    def adam_step_gather(self, w, m, v, G, gmap, lr_t, beta1, beta2, epsilon, guarded=False):
        """adam_step with the gradient read in place from the contraction's result: g_i = G.flat[gmap[i]] (gmap: device
        int32 [n], -1 = zero) -- l2o_adam_step_gather."""
        ws = self._last_ws if guarded else None
        _abi.check(self.lib.l2o_adam_step_gather(_ptr(w), _ptr(m), _ptr(v), _ptr(G), C.c_void_p(gmap.data_ptr()),
                                                 int(w.numel()), float(lr_t), float(beta1), float(beta2), float(epsilon),
                                                 None if ws is None else C.c_void_p(ws.data_ptr()), self._stream()))

    def upload(self, key, a):
        """Host array -> a PERSISTENT device tensor per key, through a pinned staging buffer with an
        asynchronous copy (the meta-training step re-uploads the packed weights after every Adam update:
        a pageable torch.as_tensor(...).to(device) of ~160 KB costs ~50 us and synchronises)."""
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        ups = self.__dict__.setdefault("_uploads", {})
        ent = ups.get(key)
        if ent is None or ent[0].numel() != a.size:
            ent = ups[key] = [self.empty(a.size), torch.empty(a.size, dtype=torch.float32).pin_memory(), None]
        dev, pin, ev = ent
        if ev is not None:
            ev.synchronize()                                 # the previous copy out of this staging buffer is done
        pin.numpy()[:] = a
        dev.copy_(pin, non_blocking=True)
        ent[2] = torch.cuda.Event()
        ent[2].record(torch.cuda.current_stream(self.device))
        return dev

    # -- state ---------------------------------------------------------------
    def state_floats(self, B, D):
        return int(self.lib.l2o_state_floats(B, D))

    def state_alloc(self, B, D):
        return self.zeros(self.state_floats(B, D))

    def state_pack(self, h1, c1, h2, c2, B, D):
        st = self.empty(self.state_floats(B, D))
        _abi.check(self.lib.l2o_state_pack(_ptr(h1), _ptr(c1), _ptr(h2), _ptr(c2), _ptr(st), B, D,
                                           self._stream()))
        return st

    def state_unpack(self, st, B, D, H=20):
        outs = [self.empty(B * D, H) for _ in range(4)]
        _abi.check(self.lib.l2o_state_unpack(_ptr(st), *[_ptr(o) for o in outs], B, D, self._stream()))
        return outs

    # -- compute -------------------------------------------------------------
    def _cprob(self, p: ProblemDesc):
        c = _abi.Problem()
        c.kind, c.B_local, c.B_global, c.D, c.M = p.kind, p.B_local, p.B_global, p.D, p.M
        c.l1, c.alpha = float(p.l1), float(p.alpha)
        c.flags = (_abi.PROB_W_SHARED if p.w_shared else 0) | \
                  (_abi.PROB_FG_TWO_PASS if _abi.get_option(_abi.OPT_FG_TWO_PASS) else 0)
        c.W, c.y, c.C, c.x_scale = _ptr(p.W), _ptr(p.y), _ptr(p.C), _ptr(p.x_scale)
        return c

    def problem_fg(self, p: ProblemDesc, x, f_part, g):
        cp = self._cprob(p)
        _abi.check(self.lib.l2o_problem_fg(C.byref(cp), _ptr(x), _ptr(f_part), _ptr(g), self._stream()))

    def problem_hvp(self, p: ProblemDesc, x, u, out):
        """out = (d g / d x) u of the analytic optimizee at x (l2o_problem_hvp), g as problem_fg returns it."""
        cp = self._cprob(p)
        scr = self.__dict__.get("_hvp_scratch")
        if scr is None or scr.numel() < p.B_local:
            scr = self._hvp_scratch = self.empty(max(p.B_local, 1))
        _abi.check(self.lib.l2o_problem_hvp(C.byref(cp), _ptr(x), _ptr(u), _ptr(out), _ptr(scr), self._stream()))

    def int_tensor(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)

    def mlp_fg(self, d: MlpDesc, indices, w1, b1, w2, b2, loss, grads):
        """grads = (gw1, gb1, gw2, gb2) device tensors or None (forward only)."""
        c = _abi.Mlp()
        c.n_in, c.n_hidden, c.n_out, c.batch = d.n_in, d.n_hidden, d.n_out, d.batch
        c.activation, c.n_data = d.activation, int(d.images.shape[0])
        c.flags = _abi.MLP_GENERIC if _abi.get_option(_abi.OPT_MLP_GENERIC) else 0
        c.images, c.labels = C.c_void_p(d.images.data_ptr()), C.c_void_p(d.labels.data_ptr())
        g = [None] * 4 if grads is None else [_ptr(t) for t in grads]
        n = int(self.lib.l2o_mlp_scratch_floats(C.byref(c)))
        if self._mlp_scratch is None or self._mlp_scratch.numel() < n:
            self._mlp_scratch = self.empty(n)
        _abi.check(self.lib.l2o_mlp_fg(C.byref(c), C.c_void_p(indices.data_ptr()), _ptr(w1), _ptr(b1), _ptr(w2),
                                       _ptr(b2), _ptr(loss), *g, _ptr(self._mlp_scratch), self._stream()))

    def _cmlp(self, d: MlpDesc):
        c = _abi.Mlp()
        c.n_in, c.n_hidden, c.n_out, c.batch = d.n_in, d.n_hidden, d.n_out, d.batch
        c.activation, c.n_data = d.activation, int(d.images.shape[0])
        c.flags = _abi.MLP_GENERIC if _abi.get_option(_abi.OPT_MLP_GENERIC) else 0
        c.images, c.labels = C.c_void_p(d.images.data_ptr()), C.c_void_p(d.labels.data_ptr())
        return c

    def mlp_deep_fg(self, d: MlpDeepDesc, indices, ws, loss, grads):
        """Loss and gradients of the MLP optimizee with SEVERAL hidden layers on ONE minibatch (l2o_mlp_deep_fg).  ws / grads:
        lists [w0, b0, w1, b1, ..., wL, bL] of device tensors (grads may be None: forward only)."""
        c = _abi.MlpDeep()
        c.n_in, c.n_out, c.batch, c.activation = d.n_in, d.n_out, d.batch, d.activation
        c.n_data, c.n_hidden_layers = int(d.images.shape[0]), len(d.hidden)
        for k, h in enumerate(d.hidden):
            c.hidden[k] = int(h)
        c.images, c.labels = C.c_void_p(d.images.data_ptr()), C.c_void_p(d.labels.data_ptr())
        n = int(self.lib.l2o_mlp_deep_scratch_floats(C.byref(c)))
        if not n:
            raise _abi.L2OUnsupported(_abi.L2O_ERR_UNSUPPORTED, "l2o_mlp_deep_fg: unsupported MLP shape %r" % (d.hidden,))
        scr = self.__dict__.get("_mlp_deep_scratch")
        if scr is None or scr.numel() < n:
            scr = self._mlp_deep_scratch = self.empty(n)
        nptr = len(ws)
        wa = (C.c_void_p * nptr)(*[t.data_ptr() for t in ws])
        ga = None if grads is None else (C.c_void_p * nptr)(*[t.data_ptr() for t in grads])
        _abi.check(self.lib.l2o_mlp_deep_fg(C.byref(c), C.c_void_p(indices.data_ptr()), wa, _ptr(loss), ga, _ptr(scr),
                                            self._stream()))

    def mlp_unroll_supported(self, spec: NetSpec, d: MlpDesc):
        """A fused persistent unroll exists for this (net, MLP optimizee) pair on this device (l2o_mlp_unroll)."""
        cc, cm = spec.to_c(), self._cmlp(d)
        return int(self.lib.l2o_mlp_unroll_supported(C.byref(cc), C.byref(cm), self._stream()))   # 2: the FAST form

    def mlp_unroll(self, spec: NetSpec, wpack, d: MlpDesc, indices, xs, sts, ms, vs, scales, T, step0, fx, hist=None):
        """T optimizer steps on the MLP optimizee in ONE launch.  indices: device int32 [T + 1, batch]; xs / sts /
        ms / vs / scales: lists of 4 device tensors (w1, b1, w2, b2; ms / vs / scales entries may be None).
        hist: None, or dict(st=, g=, m=, v=) of lists of 4 device tensors ([T, state], [T + 1, n], [T + 1, n] x 2;
        m / v None for the DM nets) that receive the history the meta-gradient needs (l2o_mlp_unroll_record)."""
        # the argument objects of a repeated launch (same buffers, same options: an evaluation loop, bench.py) are built
        # ONCE -- per call the host does a dict lookup and one ctypes call instead of ~0.1 ms of struct building
        def ptr(t):
            return 0 if t is None else t.data_ptr()
        key = (_abi.options_word(), id(d), spec.kind, spec.preprocess, tuple(spec.layers), float(spec.scale), bool(spec.tanh_output),
               float(spec.logsign_k), float(spec.beta1), float(spec.beta2), wpack.data_ptr(), indices.data_ptr(), int(T), hist is None,
               tuple(ptr(t) for ts in (xs, sts, ms, vs, scales) for t in ts))
        memo = self.__dict__.setdefault("_mlp_unroll_memo", {})
        ent = memo.get(key) if hist is None else None
        if ent is None:
            cc, cm = spec.to_c(), self._cmlp(d)
            n = int(self.lib.l2o_mlp_unroll_workspace_bytes(C.byref(cm)))
            ws = self.__dict__.get("_mlp_ws")
            if ws is None or ws.numel() < n:
                ws = self._mlp_ws = torch.zeros(n, dtype=torch.uint8, device=self.device)

            def arr(ts):
                a = (C.c_void_p * 4)()
                for k, t in enumerate(ts):
                    a[k] = None if t is None else t.data_ptr()
                return a
            ent = dict(cc=cc, cm=cm, ws=ws, arrs=(arr(xs), arr(sts), arr(ms), arr(vs), arr(scales)),
                       keep=(d, wpack, indices, list(xs), list(sts), list(ms), list(vs), list(scales)))
            if hist is None:
                if len(memo) >= 8:
                    memo.pop(next(iter(memo)))
                memo[key] = ent
        cc, cm, ws = ent["cc"], ent["cm"], ent["ws"]
        if self.__dict__.get("_mlp_ws") is not ws:           # (a larger workspace replaced it since)
            memo.pop(key, None)
            return self.mlp_unroll(spec, wpack, d, indices, xs, sts, ms, vs, scales, T, step0, fx, hist=hist)
        self._last_ws = ws
        ax, ast, am, av, asc = ent["arrs"]
        if hist is not None:
            h = _abi.MlpHist()
            for k in range(4):
                h.st[k], h.g[k] = hist["st"][k].data_ptr(), hist["g"][k].data_ptr()
                h.m[k] = None if hist.get("m") is None or hist["m"][k] is None else hist["m"][k].data_ptr()
                h.v[k] = None if hist.get("v") is None or hist["v"][k] is None else hist["v"][k].data_ptr()
            _abi.check(self.lib.l2o_mlp_unroll_record(C.byref(cc), _ptr(wpack), C.byref(cm), C.c_void_p(indices.data_ptr()),
                                                      ax, ast, am, av, asc, int(T), int(step0), _ptr(fx), C.byref(h),
                                                      C.c_void_p(ws.data_ptr()), self._stream()))
            return
        _abi.check(self.lib.l2o_mlp_unroll(C.byref(cc), _ptr(wpack), C.byref(cm), C.c_void_p(indices.data_ptr()), ax, ast,
                                           am, av, asc, int(T), int(step0), _ptr(fx), C.c_void_p(ws.data_ptr()),
                                           self._stream()))

    def mlp_unroll_multi_supported(self, spec: NetSpec, d: MlpDesc, n_inst):
        """l2o_mlp_unroll_multi applies: up to eight independent optimizee instances, one per XCD, in one launch."""
        cc, cm = spec.to_c(), self._cmlp(d)
        return bool(self.lib.l2o_mlp_unroll_multi_supported(C.byref(cc), C.byref(cm), int(n_inst), self._stream()))

    def mlp_unroll_multi(self, spec: NetSpec, wpack, d: MlpDesc, instances, T, step0):
        """T optimizer steps on up to EIGHT independent instances of the MLP optimizee in ONE launch, one instance per XCD
        (l2o_mlp_unroll_multi).  instances: list of dicts(indices=, xs=, sts=, ms=, vs=, scales=, fx=) as the arguments of
        mlp_unroll (all instances share the network, the data set and T / step0).  The argument block of a repeated
        launch (same buffers) is built once."""
        def ptr(t):
            return 0 if t is None else t.data_ptr()
        key = ("multi", _abi.options_word(), id(d), spec.kind, spec.preprocess, float(spec.scale), bool(spec.tanh_output),
               float(spec.logsign_k), float(spec.beta1), float(spec.beta2), wpack.data_ptr(), int(T),
               tuple((ptr(i["indices"]), ptr(i["fx"])) + tuple(ptr(t) for k in ("xs", "sts", "ms", "vs", "scales") for t in i[k])
                     for i in instances))
        memo = self.__dict__.setdefault("_mlp_unroll_memo", {})
        ent = memo.get(key)
        if ent is None:
            cc, cm = spec.to_c(), self._cmlp(d)
            n = int(self.lib.l2o_mlp_unroll_multi_workspace_bytes(C.byref(cm), len(instances)))
            if not n:
                raise _abi.L2OUnsupported(_abi.L2O_ERR_UNSUPPORTED, "l2o_mlp_unroll_multi: unsupported shape / instance count")
            ws = self.__dict__.get("_mlp_ws")
            if ws is None or ws.numel() < n:
                ws = self._mlp_ws = torch.zeros(n, dtype=torch.uint8, device=self.device)
            arr = (_abi.MlpInstance * len(instances))()
            for j, i in enumerate(instances):
                arr[j].indices, arr[j].fx = i["indices"].data_ptr(), i["fx"].data_ptr()
                for k in range(4):
                    arr[j].x[k], arr[j].st[k] = i["xs"][k].data_ptr(), i["sts"][k].data_ptr()
                    arr[j].m[k] = None if i["ms"][k] is None else i["ms"][k].data_ptr()
                    arr[j].v[k] = None if i["vs"][k] is None else i["vs"][k].data_ptr()
                    arr[j].x_scale[k] = None if i["scales"][k] is None else i["scales"][k].data_ptr()
            ent = dict(cc=cc, cm=cm, ws=ws, arr=arr, keep=(d, wpack, [dict(i) for i in instances]))
            if len(memo) >= 8:
                memo.pop(next(iter(memo)))
            memo[key] = ent
        ws = ent["ws"]
        if self.__dict__.get("_mlp_ws") is not ws:           # (a larger workspace replaced it since)
            memo.pop(key, None)
            return self.mlp_unroll_multi(spec, wpack, d, instances, T, step0)
        self._last_ws = ws
        _abi.check(self.lib.l2o_mlp_unroll_multi(C.byref(ent["cc"]), _ptr(wpack), C.byref(ent["cm"]), ent["arr"], len(instances),
                                                 int(T), int(step0), C.c_void_p(ws.data_ptr()), self._stream()))

    # -- prepared calls: the ctypes argument objects are built ONCE for launches that repeat with the same
    #    buffers (the T steps of a recorded unroll); per call only what changes is passed ------------------
    def prepared_mlp_fg(self, d: MlpDesc, indices, w1, b1, w2, b2, grads):
        """Returns call(loss_ptr): l2o_mlp_fg with everything but the address of the loss slot fixed."""
        c = _abi.Mlp()
        c.n_in, c.n_hidden, c.n_out, c.batch = d.n_in, d.n_hidden, d.n_out, d.batch
        c.activation, c.n_data = d.activation, int(d.images.shape[0])
        c.flags = _abi.MLP_GENERIC if _abi.get_option(_abi.OPT_MLP_GENERIC) else 0
        c.images, c.labels = C.c_void_p(d.images.data_ptr()), C.c_void_p(d.labels.data_ptr())
        g = [None] * 4 if grads is None else [_ptr(t) for t in grads]
        n = int(self.lib.l2o_mlp_scratch_floats(C.byref(c)))
        if self._mlp_scratch is None or self._mlp_scratch.numel() < n:
            self._mlp_scratch = self.empty(n)
        cref, idx0 = C.byref(c), C.c_void_p(indices.data_ptr())
        mid = (_ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2))
        tail = tuple(g) + (_ptr(self._mlp_scratch), self._stream())
        fn, check = self.lib.l2o_mlp_fg, _abi.check
        keep = (c, d, indices, w1, b1, w2, b2, grads, self._mlp_scratch)

        def call(loss_ptr, idx_ptr=None, _keep=keep):        # idx_ptr: another minibatch (device int32 [batch])
            rc = fn(cref, idx0 if idx_ptr is None else idx_ptr, *mid, loss_ptr, *tail)
            if rc:
                check(rc)
        return call

    def prepared_lstm_step_multi(self, spec: NetSpec, segs):
        """Returns call(wpack_ptr, pow1, pow2) for at most MAX_STEP_SEGS panels."""
        assert len(segs) <= self.MAX_STEP_SEGS
        arr = (_abi.StepSeg * len(segs))()
        for a, seg in zip(arr, segs):
            g, m, v, st, x, B, D = seg[:7]
            a.g, a.m, a.v, a.st, a.x, a.B, a.D = _ptr(g), _ptr(m), _ptr(v), _ptr(st), _ptr(x), B, D
            if len(seg) > 7:
                a.st_out, a.m_out, a.v_out = _ptr(seg[7]), _ptr(seg[8]), _ptr(seg[9])
        cc = spec.to_c()
        ccp, n, stream = C.byref(cc), len(segs), self._stream()
        fn, check = self.lib.l2o_cwlstm_step_multi, _abi.check
        keep = (cc, arr, segs)

        def call(wpack_ptr, pow1, pow2, _keep=keep):
            rc = fn(ccp, wpack_ptr, arr, n, pow1, pow2, stream)
            if rc:
                check(rc)
        return call

    # -- generic-`layers` nets (l2o_cwlstm_step_generic) --------------------------------
    class GenNet(object):
        """Device weights of an LSTM stack in their Sonnet layouts + the struct l2o_gen_net that points at them."""

        def __init__(self, engine, spec, params, direct=False):
            layers = tuple(int(h) for h in spec.layers)
            self.tensors = []
            c = _abi.GenNet()
            c.n_layers = len(layers)
            for l, H in enumerate(layers):
                c.hidden[l] = H
                for name, field in (("w_gates", c.w_gates), ("b_gates", c.b_gates)):
                    t = engine.tensor(params["lstm_%d" % (l + 1)][name])
                    self.tensors.append(t)
                    field[l] = t.data_ptr()
            wl, bl = engine.tensor(params["linear"]["w"]), engine.tensor(params["linear"]["b"])
            self.tensors += [wl, bl]
            c.w_lin, c.b_lin = wl.data_ptr(), bl.data_ptr()
            c.in_dim = 2 if spec.preprocess == _abi.PRE_LOGSIGN else 1
            if spec.preprocess == _abi.PRE_FC_ELU:
                wf, bf = engine.tensor(params["input_projection"]["w"]), engine.tensor(params["input_projection"]["b"])
                self.tensors += [wf, bf]
                c.w_fc, c.b_fc = wf.data_ptr(), bf.data_ptr()
                c.in_dim = int(wf.shape[1])
            c.direct_inputs = 1 if direct else 0
            self.c, self.layers = c, layers

    def gen_net(self, spec: NetSpec, params: dict, direct=False):
        return HipEngine.GenNet(self, spec, params, direct)

    def gen_state_alloc(self, gen, N):
        return self.zeros(int(self.lib.l2o_gen_state_floats(C.byref(gen.c), int(N))))

    def lstm_step_generic(self, spec: NetSpec, gen, g, m_tilde, m, v, pow1, pow2, st, x, N):
        cc = spec.to_c()
        _abi.check(self.lib.l2o_cwlstm_step_generic(C.byref(cc), C.byref(gen.c), _ptr(g), _ptr(m_tilde), _ptr(m), _ptr(v),
                                                    float(pow1), float(pow2), _ptr(st), _ptr(x), int(N), self._stream()))

    def bwd_step_generic(self, spec: NetSpec, gen, io: dict, pow1, pow2, N):
        """One BPTT step of an ANY-`layers` stack (l2o_cwlstm_bwd_step_generic).  io: device tensors keyed by the fields
        of struct l2o_gen_bwd_io (act / dz: lists, one tensor per layer; missing = NULL)."""
        c = _abi.GenBwdIO()
        for k, _ in _abi.GenBwdIO._fields_:
            if k in ("act", "dz"):
                for l, t in enumerate(io[k]):
                    getattr(c, k)[l] = t.data_ptr()
            else:
                t = io.get(k)
                setattr(c, k, None if t is None else t.data_ptr())
        cc = spec.to_c()
        _abi.check(self.lib.l2o_cwlstm_bwd_step_generic(C.byref(cc), C.byref(gen.c), C.byref(c), float(pow1), float(pow2),
                                                        int(N), self._stream()))

    def lstm_step(self, spec: NetSpec, wpack, g, m, v, pow1, pow2, st, x, B, D):
        if isinstance(wpack, HipEngine.GenNet):
            return self.lstm_step_generic(spec, wpack, g, None, m, v, pow1, pow2, st, x, B * D)
        cc = spec.to_c()
        _abi.check(self.lib.l2o_cwlstm_step(C.byref(cc), _ptr(wpack), _ptr(g), _ptr(m), _ptr(v),
                                            float(pow1), float(pow2), _ptr(st), _ptr(x), B, D,
                                            self._stream()))

    MAX_STEP_SEGS = 8

    def lstm_step_multi(self, spec: NetSpec, wpack, segs, pow1, pow2):
        """One launch for several variables that share a network.  segs: list of
        (g, m, v, st, x, B, D[, st_out, m_out, v_out]) with device tensors (m, v, st may be None)."""
        if isinstance(wpack, HipEngine.GenNet):              # generic-`layers` net: one launch per variable
            for seg in segs:
                g, m, v, st, x, B, D = seg[:7]
                if len(seg) > 7 and seg[7] is not None:      # history chaining: the step runs in place on the copies
                    st_out, m_out, v_out = seg[7:10]
                    st_out.copy_(st)
                    if m is not None:
                        m_out.copy_(m); v_out.copy_(v)
                        m, v = m_out, v_out
                    st = st_out
                self.lstm_step_generic(spec, wpack, g, None, m, v, pow1, pow2, st, x, B * D)
            return
        for i in range(0, len(segs), self.MAX_STEP_SEGS):
            chunk = segs[i:i + self.MAX_STEP_SEGS]
            arr = (_abi.StepSeg * len(chunk))()
            for a, seg in zip(arr, chunk):
                g, m, v, st, x, B, D = seg[:7]
                a.g, a.m, a.v, a.st, a.x, a.B, a.D = _ptr(g), _ptr(m), _ptr(v), _ptr(st), _ptr(x), B, D
                if len(seg) > 7:                             # (st_out, m_out, v_out): out of place, see l2o_step_seg
                    a.st_out, a.m_out, a.v_out = _ptr(seg[7]), _ptr(seg[8]), _ptr(seg[9])
            cc = spec.to_c()
            _abi.check(self.lib.l2o_cwlstm_step_multi(C.byref(cc), _ptr(wpack), arr, len(chunk), float(pow1),
                                                      float(pow2), self._stream()))

    def bwd_step(self, spec: NetSpec, weights: dict, io: dict, pow1, pow2, B, D):
        """One BPTT step (l2o_cwlstm_bwd_step).  weights / io: dicts of device tensors keyed by the
        field names of struct l2o_net_weights / l2o_bwd_io (missing = NULL)."""
        cc = spec.to_c()
        w = _abi.NetWeights()
        for k, _ in _abi.NetWeights._fields_:
            setattr(w, k, None if weights.get(k) is None else weights[k].data_ptr())
        b = _abi.BwdIO()
        for k, _ in _abi.BwdIO._fields_:
            if k in ("a_stride", "b_stride"):
                setattr(b, k, int(io.get(k) or 0))
            elif k == "dg":
                b.dg = None if io.get("dg") is None else io["dg"].data_ptr()
            else:
                setattr(b, k, None if io.get(k) is None else io[k].data_ptr())
        _abi.check(self.lib.l2o_cwlstm_bwd_step(C.byref(cc), C.byref(w), C.byref(b), float(pow1), float(pow2),
                                                B, D, self._stream()))

    def bwd_multi(self, spec: NetSpec, weights: dict, segs, carry_in, carry_out, A, Bm, pow1, pow2):
        """One BPTT step for several panels of one network in one launch (l2o_cwlstm_bwd_multi).
        segs: list of dict(g=, m=, v=, st_prev=, dx_next=, B=, D=); A [rows, KA], Bm [rows, KB] and the
        carries [4, rows, 20] are shared, rows = 16 * (total tiles)."""
        cc = spec.to_c()
        w = _abi.NetWeights()
        for k, _ in _abi.NetWeights._fields_:
            setattr(w, k, None if weights.get(k) is None else weights[k].data_ptr())
        arr = (_abi.BwdSeg * len(segs))()
        for a, sg in zip(arr, segs):
            for k in ("g", "m", "v", "st_prev", "dx_next"):
                setattr(a, k, None if sg.get(k) is None else sg[k].data_ptr())
            a.B, a.D = int(sg["B"]), int(sg["D"])
        _abi.check(self.lib.l2o_cwlstm_bwd_multi(C.byref(cc), C.byref(w), arr, len(segs), _ptr(carry_in),
                                                 _ptr(carry_out), _ptr(A), _ptr(Bm), float(pow1), float(pow2),
                                                 self._stream()))

    bwd_unroll_any_d = True       # l2o_cwlstm_bwd_unroll takes panels of any D (per-problem tiles, ragged last tile)

    bwd_unroll_compact = True     # l2o_cwlstm_bwd_unroll_compact / l2o_cwlstm_wgrad_compact exist (ABI v12)

    def bwd_unroll(self, spec: NetSpec, weights: dict, panels, T, step0, A, Bm, carry_in=None, carry_out=None,
                   table=None, compact=False):
        """All T BPTT steps of the panels that share one network in one launch (l2o_cwlstm_bwd_unroll).
        panels: list of dict(B=, D=, gs=[T tensors], sts=[T], ms=[T] | None, vs=, dxs=[T] | None,
        g_final= tensor | None); A [T, rows, KA], Bm [T, rows, KB], rows = 16 * sum_panels B * ceil(D / 16).
        compact: A is [T + 1, rows, KA - 40] -- the rows without their duplicated h1(t-1) / h2(t-1) columns
        (l2o_cwlstm_bwd_unroll_compact; read by wgrad_compact)."""
        cc = spec.to_c()
        w = _abi.NetWeights()
        for k, _ in _abi.NetWeights._fields_:
            setattr(w, k, None if weights.get(k) is None else weights[k].data_ptr())
        if table is None:
            table = self.bwd_table(panels, T)
        arr = (_abi.BwdUnrollSeg * len(panels))()
        for a, pn in zip(arr, panels):
            a.B, a.D = int(pn["B"]), int(pn["D"])
            a.g_final = None if pn.get("g_final") is None else pn["g_final"].data_ptr()
        fn = self.lib.l2o_cwlstm_bwd_unroll_compact if compact else self.lib.l2o_cwlstm_bwd_unroll
        _abi.check(fn(C.byref(cc), C.byref(w), arr, len(panels), C.c_void_p(table.data_ptr()), int(T),
                      int(step0), _ptr(carry_in), _ptr(carry_out), _ptr(A), _ptr(Bm), self._stream()))
        self._bwd_table = table                              # keep the pointer table alive until the stream has run the kernel

    def bwd_table(self, panels, T):
        """The device pointer table [T][len(panels)][5] of l2o_cwlstm_bwd_unroll (g, m, v, st_prev, dx_next)."""
        ptr = lambda x: 0 if x is None else x.data_ptr()
        rows = []
        for t in range(T):
            for pn in panels:
                rows.append([ptr(pn["gs"][t]), ptr(pn["ms"][t]) if pn.get("ms") else 0, ptr(pn["vs"][t]) if pn.get("vs") else 0,
                             ptr(pn["sts"][t]), ptr(pn["dxs"][t]) if pn.get("dxs") else 0])
        return torch.tensor(rows, dtype=torch.int64).to(self.device, non_blocking=False)

    def unroll_supported(self, spec: NetSpec, p: ProblemDesc, record=False):
        """A fused kernel exists for the pair; record=True: one that also records the BPTT history."""
        cc, cp = spec.to_c(), self._cprob(p)
        fn = self.lib.l2o_unroll_record_supported if record else self.lib.l2o_unroll_supported
        return bool(fn(C.byref(cc), C.byref(cp)))

    def unroll(self, spec: NetSpec, wpack, p: ProblemDesc, x, st, m, v, T, step0, fx_part, hist=None, fx=None, x0=None,
               zero_state=False):
        """hist: None, or dict(st=[T, state_floats], g=[T, B*D], m=, v= (RNNProp), g_final=[B*D]) of
        device tensors that receive the per-step history the meta-gradient needs (l2o_unroll_record).
        fx: None, or a device tensor [T + 1] that receives the batch-mean loss of every step (l2o_unroll_reduce:
        the reduction rides in the unroll's epilogue kernel where there is one).  x0 / zero_state (with fx): start
        from x0 and the zero LSTM state / moments instead of the contents of x / st / m / v (`reset` folded in)."""
        assert fx is not None or (x0 is None and not zero_state)
        cc, cp = spec.to_c(), self._cprob(p)
        nbytes = int(self.lib.l2o_unroll_workspace_bytes(C.byref(cc), C.byref(cp), int(T)))
        ws = None
        if nbytes:
            ws = self._workspace
            layout = int(self.lib.l2o_unroll_workspace_layout(C.byref(cc), C.byref(cp)))
            if ws is None or ws.numel() < nbytes:
                ws = self._workspace = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            elif layout != self.__dict__.get("_ws_layout"):
                # the granule area moved: zero the workspace once (the library keeps it clean between launches)
                _abi.check(self.lib.l2o_unroll_workspace_init(C.c_void_p(ws.data_ptr()), ws.numel(), self._stream()))
            self._ws_layout = layout
        self._last_ws = ws
        wsp = None if ws is None else C.c_void_p(ws.data_ptr())
        flags = _abi.UNROLL_ZERO_STATE if zero_state else 0
        h = None
        if hist is not None:
            h = _abi.UnrollHist()
            for k, _ in _abi.UnrollHist._fields_:
                setattr(h, k, None if hist.get(k) is None else hist[k].data_ptr())
        if fx is not None:
            _abi.check(self.lib.l2o_unroll_reduce(C.byref(cc), _ptr(wpack), C.byref(cp), _ptr(x0), _ptr(x), _ptr(st), _ptr(m),
                                                  _ptr(v), int(T), int(step0), flags, _ptr(fx_part), _ptr(fx),
                                                  wsp, None if h is None else C.byref(h), self._stream()))
        elif hist is None:
            _abi.check(self.lib.l2o_unroll(C.byref(cc), _ptr(wpack), C.byref(cp), _ptr(x), _ptr(st), _ptr(m),
                                           _ptr(v), int(T), int(step0), _ptr(fx_part), wsp, self._stream()))
        else:
            _abi.check(self.lib.l2o_unroll_record(C.byref(cc), _ptr(wpack), C.byref(cp), _ptr(x), _ptr(st), _ptr(m),
                                                  _ptr(v), int(T), int(step0), _ptr(fx_part), wsp, C.byref(h),
                                                  self._stream()))

    def prepared_unroll(self, spec: NetSpec, wpack, p: ProblemDesc, x, st, m, v, T, fx_part, x0, zero_state):
        """l2o_unroll_reduce with every argument object built ONCE: returns call(fx, step0) for launches that repeat with
        the same buffers (an evaluation loop that re-runs one problem instance from x0; bench.py's ring of instances) --
        per call the host does one ctypes call (~10 us) instead of rebuilding the structs (~0.2 ms, which is a whole
        config-2 unroll).  Returns None when the launch needs the general path (a workspace that must be (re)initialised).
        The closure is valid while the engine's workspace AND its layout
        are unchanged (it checks both and returns False otherwise) and the option settings are the ones it was built
        under (the caller keys its cache on them)."""
        cc, cp = spec.to_c(), self._cprob(p)
        nbytes = int(self.lib.l2o_unroll_workspace_bytes(C.byref(cc), C.byref(cp), int(T)))
        ws, layout = None, None
        if nbytes:
            ws = self._workspace
            layout = int(self.lib.l2o_unroll_workspace_layout(C.byref(cc), C.byref(cp)))
            if ws is None or ws.numel() < nbytes or layout != self.__dict__.get("_ws_layout"):
                return None                                  # unroll() allocates / re-initialises it first
        wsp = None if ws is None else C.c_void_p(ws.data_ptr())
        flags = _abi.UNROLL_ZERO_STATE if zero_state else 0
        args_a = (C.byref(cc), _ptr(wpack), C.byref(cp), _ptr(x0), _ptr(x), _ptr(st), _ptr(m), _ptr(v), int(T))
        args_b = (flags, _ptr(fx_part))
        fn, check, stream = self.lib.l2o_unroll_reduce, _abi.check, self._stream
        keep = (cc, cp, wpack, p, x, st, m, v, fx_part, x0, ws)
        eng = self

        def call(fx, step0=1, _keep=keep):
            if eng._workspace is not ws or (ws is not None and eng.__dict__.get("_ws_layout") != layout):
                # the workspace was replaced, or another graph / problem size re-initialised it for a different
                # (B_local, CH) layout since this call was prepared (its granule area may now overlap
                # that layout's loss partials -- stale tags): the caller rebuilds through unroll()
                return False
            rc = fn(*args_a, int(step0), *args_b, C.c_void_p(fx.data_ptr()), wsp, None, stream())
            if rc:
                check(rc)
            eng._last_ws = ws
            return True
        return call

    def prefetch_unroll_status(self):
        """Queue the copy of the status word into pinned host memory on the launch stream, so that the host sync the
        caller is about to make anyway (reading the losses back) delivers it too: check_unroll_status() then costs
        no round trip of its own."""
        ws = self._last_ws
        if ws is None:
            return
        pin = self.__dict__.get("_status_pin")
        if pin is None:
            pin = self._status_pin = torch.zeros(4, dtype=torch.uint8).pin_memory()
        pin.copy_(ws[:4], non_blocking=True)
        self._status_pending = ws

    def last_unroll_form(self):
        """(kernel name, dispatches) of the last fused unroll this thread launched (l2o_last_unroll_form)."""
        return _abi.last_unroll_form()

    def last_unroll_exchanges(self):
        """The last fused launch ran a kernel whose workgroups wait for partner workgroups (the two-CU unroll, the
        persistent MLP unroll): the only launches that can end in L2OPartnerTimeout."""
        return (int(self.lib.l2o_last_unroll_form()) & 0xff) in _abi.FORMS_WITH_EXCHANGE

    def last_loop_ticks(self):
        """(step-loop cycles, kernel-entry-to-exit cycles) of wave 0 of workgroup 0 in the last fused launch -- shader-clock
        cycles (s_memtime), workspace bytes 16..31; the two-CU kernel, k_unroll_lds, l2o_mlp_unroll -- or None.
        Synchronises the host."""
        ws = self._last_ws
        if ws is None:
            return None
        loop, total = (int(v) for v in ws[16:32].view(torch.int64).cpu())
        return (loop, total) if loop > 0 and total >= loop else None

    def unroll_status_tensor(self):
        """The sticky status word of the last fused launch's workspace as a device int32 [1] view (None without a
        workspace): what a sharded training step all-reduces (MAX) ahead of its guarded meta-step, so that every rank
        skips the update when any rank's unroll failed."""
        ws = self._last_ws
        return None if ws is None else ws[:4].view(torch.int32)

    def inject_unroll_fault(self, on=True):
        """TEST HOOK (include/l2o_abi.h: workspace bytes 8..11): while set, every launch of an exchanging kernel on the
        engine's workspaces behaves as if its partners never showed up.  Cleared by check_unroll_status / by hand."""
        for ws in (self._workspace, self.__dict__.get("_mlp_ws")):
            if ws is not None:
                ws[8:12].view(torch.int32).fill_(1 if on else 0)

    def check_unroll_status(self):
        """After a host sync: raise if the split-problem kernel reported a partner timeout (_abi.L2OPartnerTimeout;
        the status word -- and the test hook's fault word -- are cleared: the caller handles it)."""
        ws = self._last_ws
        if ws is not None:
            if self.__dict__.pop("_status_pending", None) is ws:
                hdr = self._status_pin.numpy().tobytes()      # (prefetched ahead of the sync that just happened)
            else:
                hdr = ws[:4].cpu().numpy().tobytes()
            if hdr != b"\0\0\0\0":
                # The status word is sticky: handled here, cleared here -- together with the WHOLE workspace:
                # a launch that timed out may have left granules tagged for steps it never completed, and the persistent
                # MLP kernel advances its tag salt from inside the kernel, so a dead launch can leave valid-looking tags
                # for the next one.  Zero memory is the initial state of both workspaces (l2o_unroll_workspace_init);
                # this also clears the test hook's one-shot fault word (bytes 8..11).
                ws.zero_()
            _abi.check(self.lib.l2o_unroll_status(hdr))

    def atb(self, A, B):
        """A^T B for tall-skinny fp32 device matrices A [R, KA], B [R, KB] (l2o_atb: split-K over the rows on the
        matrix cores, fixed-order reduction) -> new [KA, KB] device tensor."""
        R, KA = A.shape
        KB = B.shape[1]
        n = int(self.lib.l2o_atb_workspace_bytes(int(R), int(KA), int(KB)))
        ws = self.__dict__.get("_atb_ws")
        if ws is None or ws.numel() * 4 < n:
            ws = self._atb_ws = self.empty((n + 3) // 4)
        out = self.empty(KA, KB)
        _abi.check(self.lib.l2o_atb(_ptr(A), _ptr(B), int(R), int(KA), int(KB), _ptr(out), _ptr(ws), self._stream()))
        return out

    def wgrad(self, spec: NetSpec, A, B):
        """The weight-gradient blocks of A^T B for the BPTT rows of a layers=(20,20) net (l2o_cwlstm_wgrad: only the
        tiles that are somebody's gradient; zeros elsewhere) -> new [KA, KB] device tensor."""
        R, KA = A.shape
        KB = B.shape[1]
        cc = spec.to_c()
        ka, kb = C.c_int32(0), C.c_int32(0)
        _abi.check(self.lib.l2o_cwlstm_wgrad_dims(C.byref(cc), C.byref(ka), C.byref(kb)))
        assert (ka.value, kb.value) == (KA, KB), ((ka.value, kb.value), (KA, KB))
        n = int(self.lib.l2o_atb_workspace_bytes(int(R), int(KA), int(KB)))
        ws = self.__dict__.get("_atb_ws")
        if ws is None or ws.numel() * 4 < n:
            ws = self._atb_ws = self.empty((n + 3) // 4)
        out = self.empty(KA, KB)
        _abi.check(self.lib.l2o_cwlstm_wgrad(C.byref(cc), _ptr(A), _ptr(B), int(R), _ptr(out), _ptr(ws), self._stream()))
        return out

    def wgrad_compact(self, spec: NetSpec, Ac, Bm):
        """wgrad for the compact rows of bwd_unroll(compact=True): Ac [T + 1, rows, KA - 40], Bm [T, rows, KB] -> [KA, KB]
        (l2o_cwlstm_wgrad_compact)."""
        T, R, KB = Bm.shape
        cc = spec.to_c()
        ka, kb = C.c_int32(0), C.c_int32(0)
        _abi.check(self.lib.l2o_cwlstm_wgrad_dims(C.byref(cc), C.byref(ka), C.byref(kb)))
        KA = ka.value
        assert kb.value == KB and tuple(Ac.shape) == (T + 1, R, KA - 40), (tuple(Ac.shape), (T + 1, R, KA - 40), KB)
        n = int(self.lib.l2o_atb_workspace_bytes(int(T * R), int(KA), int(KB)))
        ws = self.__dict__.get("_atb_ws")
        if ws is None or ws.numel() * 4 < n:
            ws = self._atb_ws = self.empty((n + 3) // 4)
        out = self.empty(KA, KB)
        _abi.check(self.lib.l2o_cwlstm_wgrad_compact(C.byref(cc), _ptr(Ac), _ptr(Bm), int(T), int(R), _ptr(out), _ptr(ws),
                                                     self._stream()))
        return out

    # -- small vector passes of the meta-gradient (ABI v11, csrc/l2o_vecops.h) -------------------------------
    def suffix_sums(self, gs, g_final, out):
        """out[t] = g_final + sum_{tau > t} gs[tau] (gs: list of T device tensors of n floats; out: [T, n])."""
        T, n = len(gs), g_final.numel()
        if T == 0:
            return out
        # the pointer table: cached on the addresses it holds (a planned unroll records into the same history buffers
        # every step -- one upload for the whole training run), uploaded from a pinned staging buffer otherwise: a
        # pageable H2D copy synchronises the host with the stream on every BPTT
        ptrs = tuple(g.data_ptr() for g in gs)
        cache = self.__dict__.setdefault("_suffix_tables", {})
        table = cache.get(ptrs)
        if table is None:
            if len(cache) >= 16:
                cache.clear()
            pin = torch.tensor(ptrs, dtype=torch.int64).pin_memory()
            table = torch.empty(T, dtype=torch.int64, device=self.device)
            table.copy_(pin, non_blocking=True)
            cache[ptrs] = table
            self._suffix_pin = pin                       # (the staging buffer outlives the asynchronous copy)
        _abi.check(self.lib.l2o_suffix_sums(C.c_void_p(table.data_ptr()), _ptr(g_final), _ptr(out), n, T, self._stream()))
        return out

    def colsum(self, A, out=None, accumulate=False):
        """Column sums of a [rows, cols] (or batched [b, rows, cols]) device matrix, fixed summation order."""
        A3 = A if A.dim() == 3 else A.unsqueeze(0)
        assert A3.is_contiguous()
        b, rows, cols = A3.shape
        if out is None:
            out = self.empty(b, cols) if A.dim() == 3 else self.empty(cols)
        n = int(self.lib.l2o_colsum_scratch_floats(b, cols))
        scr = self.__dict__.get("_colsum_scratch")
        if scr is None or scr.numel() < n:
            scr = self._colsum_scratch = self.empty(n)
        _abi.check(self.lib.l2o_colsum(_ptr(A3), b, rows, cols, _ptr(out), 1 if accumulate else 0, _ptr(scr), self._stream()))
        return out

    def lincomb(self, out, a, ca=1.0, b=None, cb=0.0, c=None, cc=0.0):
        """out = ca a + cb b + cc c (elementwise; out may alias an input)."""
        _abi.check(self.lib.l2o_lincomb(_ptr(out), _ptr(a), float(ca), _ptr(b), float(cb), _ptr(c), float(cc), out.numel(),
                                        self._stream()))
        return out

    def rnnprop_input_adjoint(self, Bm, du_col, H, w_fc, g, m, v, pow1, pow2, beta1, beta2, dm, dv, dg):
        """second_derivatives for RNNProp: dg = dL/dg_t through (m~, g~) and the moment recurrences; dm, dv in place."""
        n = g.numel()
        _abi.check(self.lib.l2o_rnnprop_input_adjoint(_ptr(Bm), int(Bm.stride(0)), int(du_col), int(H), _ptr(w_fc), _ptr(g),
                                                      _ptr(m), _ptr(v), float(pow1), float(pow2), float(beta1), float(beta2),
                                                      _ptr(dm), _ptr(dv), _ptr(dg), n, self._stream()))

    def reduce_fx(self, fx_part, T1, B_local, B_global, fx):
        _abi.check(self.lib.l2o_reduce_fx(_ptr(fx_part), int(T1), int(B_local), int(B_global), _ptr(fx),
                                          self._stream()))

    def synchronize(self):
        torch.cuda.synchronize(self.device)


def pack_weights_host(lib, spec: NetSpec, params: dict):
    """Host-only re-layout of a ``.l2l`` dict into MFMA-fragment order
    (l2o_wpack_host).  Usable without a GPU."""
    cc = spec.to_c()
    n = int(lib.l2o_wpack_floats(C.byref(cc)))
    if n == 0:
        raise _abi.L2OUnsupported(
            _abi.L2O_ERR_UNSUPPORTED,
            "no HIP kernel for an optimizer net with layers=%r (implemented: (20, 20) and ())" % (spec.layers,))
    out = np.zeros((n,), np.float32)

    def arr(mod, var):
        if mod not in params:
            return None
        return np.ascontiguousarray(params[mod][var], dtype=np.float32)

    keep = [arr("lstm_1", "w_gates"), arr("lstm_1", "b_gates"), arr("lstm_2", "w_gates"),
            arr("lstm_2", "b_gates"), arr("linear", "w"), arr("linear", "b"),
            arr("input_projection", "w"), arr("input_projection", "b")]
    ptrs = [None if a is None else a.ctypes.data_as(C.c_void_p) for a in keep]
    _abi.check(lib.l2o_wpack_host(C.byref(cc), *ptrs, out.ctypes.data_as(C.c_void_p)))
    return out


_default_engine = None


def default_engine():
    """The process-wide HipEngine (created on first use; raises without GPU/extension)."""
    global _default_engine
    if _default_engine is None:
        _default_engine = HipEngine()
    return _default_engine


def set_default_engine(engine):
    global _default_engine
    _default_engine = engine
