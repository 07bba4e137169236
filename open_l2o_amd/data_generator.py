"""Imitation-learning data: trajectories of analytic optimizers on the optimizee.

The reference's ``DM/data_generator.py`` (``data_loader``): run Adam / RMSProp / Nesterov
momentum (all lr 0.01, TF-1.x update rules) on the optimizee and record, per unroll of
``unroll_len`` steps and per subset of variables, the gradients fed to the optimizer
("inputs", [T, P]) and the updates it made ("labels", [T, P]).  ``MetaOptimizer``'s ``mt``
unrolls (``meta.MtUnroll``) regress the learned optimizer onto them.

Gradients come from the same HIP kernels as the unroll (``UnrollGraph.gradients``); the
optimizers' few elementwise updates are torch tensor ops on the device.
"""
import numpy as np
import torch


class _Adam(object):              # tf.train.AdamOptimizer(0.01): DM/data_generator.py:48-50
    def __init__(self, lr=0.01, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps

    def reset(self, xs):
        self.t = 0
        self.m = [torch.zeros_like(x) for x in xs]
        self.v = [torch.zeros_like(x) for x in xs]

    def apply(self, xs, gs):
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for x, g, m, v in zip(xs, gs, self.m, self.v):
            m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            x.addcdiv_(m, v.sqrt().add_(self.eps), value=-float(lr_t))


class _RMSProp(object):           # tf.train.RMSPropOptimizer(0.01): decay 0.9, momentum 0, eps 1e-10, rms slot = 1
    def __init__(self, lr=0.01, decay=0.9, momentum=0.0, eps=1e-10):
        self.lr, self.decay, self.momentum, self.eps = lr, decay, momentum, eps

    def reset(self, xs):
        self.ms = [torch.ones_like(x) for x in xs]
        self.mom = [torch.zeros_like(x) for x in xs]

    def apply(self, xs, gs):
        for x, g, ms, mom in zip(xs, gs, self.ms, self.mom):
            ms.mul_(self.decay).addcmul_(g, g, value=1.0 - self.decay)
            mom.mul_(self.momentum).add_(g / (ms + self.eps).sqrt(), alpha=self.lr)
            x.sub_(mom)


class _Nag(object):               # tf.train.MomentumOptimizer(0.01, 0.9, use_nesterov=True)
    def __init__(self, lr=0.01, momentum=0.9):
        self.lr, self.momentum = lr, momentum

    def reset(self, xs):
        self.acc = [torch.zeros_like(x) for x in xs]

    def apply(self, xs, gs):
        for x, g, acc in zip(xs, gs, self.acc):
            acc.mul_(self.momentum).add_(g)
            x.sub_(g + self.momentum * acc, alpha=self.lr)


_OPTIMIZERS = {"adam": _Adam, "rmsprop": _RMSProp, "nag": _Nag}


class data_loader(object):
    """``data_loader(make_loss, x, constants, subsets, scale, optimizers, unroll_len)``.
    DM/data_generator.py:35-62.  ``x`` / ``scale`` are the lists ``meta_minimize`` returned."""

    def __init__(self, make_loss, x, constants, subsets, scale, optimizers, unroll_len):
        self.unroll_len = unroll_len
        self.optimizers = optimizers.split(",")
        for name in self.optimizers:
            if name not in _OPTIMIZERS:
                raise ValueError("unknown optimizer %r (adam, rmsprop, nag)" % (name,))
        self.num_subsets = len(subsets)
        self.subsets = subsets
        self.x = x
        self.scale = scale
        self.graph = x[0]._graph

    def _flat(self, tensors):
        """[P] host vector per subset: flattened variables of the subset concatenated."""
        eng = self.graph.engine
        return [np.concatenate([eng.to_numpy(tensors[i]).reshape(-1) for i in subset]) for subset in self.subsets]

    def get_data(self, task_i, sess, num_unrolls, assign_func, rd_scale_bound, if_scale=True, mt_k=1):
        """DM/data_generator.py:72-124: {"inputs": [unroll][subset] -> [T, P], "labels": likewise}."""
        graph = self.graph
        opt = _OPTIMIZERS[self.optimizers[task_i]]()
        graph.reset()                                       # reset_x: fresh x and problem data
        if if_scale:
            from . import meta
            r_scale = [meta.synced_scale(k.shape, rd_scale_bound) for k in self.scale]   # same factors on every rank
            feed_rs = {p: v for p, v in zip(self.scale, r_scale)}
            assert assign_func is not None
            assign_func([sess.run(v) / meta.local_slice(v, r) for v, r in zip(self.x, r_scale)])
        else:
            feed_rs = {}
        xs = [v.value for v in self.x]
        opt.reset(xs)
        data = {"inputs": [], "labels": []}
        x_prev = self._flat(xs)
        for _ in range(num_unrolls):
            inputs, labels = [], []
            for _ in range(self.unroll_len):
                gs = graph.gradients(feed_rs)
                inputs.append(self._flat(gs))
                opt.apply(xs, [g.view_as(x) for g, x in zip(gs, xs)])
                for _ in range(mt_k - 1):
                    gk = graph.gradients(feed_rs)
                    opt.apply(xs, [g.view_as(x) for g, x in zip(gk, xs)])
                x_cur = self._flat(xs)
                labels.append([cur - prev for cur, prev in zip(x_cur, x_prev)])
                x_prev = x_cur
            data["inputs"].append([np.stack([ipt[i] for ipt in inputs]) for i in range(self.num_subsets)])
            data["labels"].append([np.stack([lb[i] for lb in labels]) for i in range(self.num_subsets)])
        return data
