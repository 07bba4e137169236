"""UnrollGraph's meta-step (split out of meta.py in round 4): TF-1.x Adam on the optimizer networks' weights -- on the
device (l2o_adam_step / _gather / _guarded + l2o_wpack_device) for the LSTM nets, NumPy for the rest."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _abi, networks
from ._graph_core import PackedState, _DevGrad, _LazyHost, _term_vars, _world, rng  # noqa: F401


class AdamMixin(object):
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
                and all(sr[0] is srcs[0][0] for sr in srcs)):
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

