"""RNNProp evaluation meta-optimizer -- the reference's ``DM/meta_rnnprop_eval.py`` API.

``MetaOptimizer(beta1, beta2, **net_config)``; ``meta_loss`` returns
``(MetaLoss, scale, x, step)`` (DM/meta_rnnprop_eval.py:466) where ``scale`` are the
per-variable x-scale placeholders, ``x`` the optimizee variables and ``step`` the
placeholder the harness feeds with ``i * unroll_len + 1`` (DM/util.py:59-60, 85-86).
The Adam moments m, v (DM/meta_rnnprop_eval.py time_step/update) are carried by
``update`` and zeroed by ``reset``; the inputs (m~, g~) are formed inside the kernels.
"""
from . import meta as _meta
from .meta import MetaLoss, MetaStep, set_random_seed  # noqa: F401


class MetaOptimizer(_meta.MetaOptimizer):
    _rnnprop = True

    def __init__(self, beta1, beta2, **kwargs):
        super(MetaOptimizer, self).__init__(**kwargs)
        self.beta1 = beta1
        self.beta2 = beta2

    def meta_loss(self, make_loss, len_unroll, net_assignments=None, second_derivatives=False):
        graph = self._build_graph(make_loss, len_unroll, net_assignments, second_derivatives)
        return self._handles(graph), graph.scale, graph.x, graph.step

    def meta_minimize(self, make_loss, len_unroll, learning_rate=0.01, **kwargs):
        """DM/meta_rnnprop_eval.py:468-485: (MetaStep, scale, x, seq_step)."""
        info, scale, x, seq_step = self.meta_loss(make_loss, len_unroll, **kwargs)
        self._graph.learning_rate = learning_rate
        return _meta.MetaStep(_meta.Fetch(self._graph, "step"), *info[1:]), scale, x, seq_step
