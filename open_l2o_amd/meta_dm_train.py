"""L2O-DM training meta-optimizer -- the reference's ``DM/meta_dm_train.py`` API: forward
unroll with the per-variable x-scale placeholders (DM/meta_dm_train.py:336-338, 384, 415), the
Adam meta-step, and ``num_mt`` imitation-learning ("mt") unrolls (:421-499).

``MetaOptimizer(num_mt, **net_config)``; ``meta_loss`` returns the reference's 10-tuple
(DM/meta_dm_train.py:526-527): ``(MetaLoss, scale, x, constants, subsets, loss_mt,
update_mt, reset_mt, mt_labels, mt_inputs)``.
"""
from . import meta as _meta
from .meta import MetaLoss, MetaStep, set_random_seed  # noqa: F401


class MetaOptimizer(_meta.MetaOptimizer):
    def __init__(self, num_mt, **kwargs):
        super(MetaOptimizer, self).__init__(**kwargs)
        self.num_mt = int(num_mt)

    def meta_loss(self, make_loss, len_unroll, net_assignments=None, second_derivatives=False):
        graph = self._build_graph(make_loss, len_unroll, net_assignments, second_derivatives)
        loss_mt, _, update_mt, reset_mt, mt_labels, mt_inputs = _meta.make_mt_handles(graph, self.num_mt)
        return (self._handles(graph), graph.scale, graph.x, graph.constants, graph.subsets,
                loss_mt, update_mt, reset_mt, mt_labels, mt_inputs)

    def meta_minimize(self, make_loss, len_unroll, learning_rate=0.01, **kwargs):
        """DM/meta_dm_train.py:529-558: (MetaStep, scale, x, constants, subsets, loss_mt, steps_mt,
        update_mt, reset_mt, mt_labels, mt_inputs); every mt task has its own Adam optimizer."""
        out = self.meta_loss(make_loss, len_unroll, **kwargs)
        graph = self._graph
        graph.learning_rate = learning_rate
        loss_mt, steps_mt, update_mt, reset_mt, mt_labels, mt_inputs = _meta.make_mt_handles(graph, self.num_mt)
        for m in graph.mt:
            m.learning_rate = learning_rate
        step = _meta.MetaStep(_meta.Fetch(graph, "step"), *out[0][1:])
        return (step,) + tuple(out[1:5]) + (loss_mt, steps_mt, update_mt, reset_mt, mt_labels, mt_inputs)
