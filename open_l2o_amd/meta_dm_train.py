"""L2O-DM training meta-optimizer -- the reference's ``DM/meta_dm_train.py`` API (forward
unroll with the per-variable x-scale placeholders, DM/meta_dm_train.py:336-338, 384, 415;
the imitation "mt" unrolls and the Adam meta-step need the meta-gradient).

``MetaOptimizer(num_mt, **net_config)``; ``meta_loss`` returns the reference's 10-tuple
(DM/meta_dm_train.py:526-527): ``(MetaLoss, scale, x, constants, subsets, loss_mt,
update_mt, reset_mt, mt_labels, mt_inputs)`` with empty mt lists for ``num_mt == 0``.
"""
from . import meta as _meta
from .meta import MetaLoss, MetaStep, set_random_seed  # noqa: F401


class MetaOptimizer(_meta.MetaOptimizer):
    def __init__(self, num_mt, **kwargs):
        super(MetaOptimizer, self).__init__(**kwargs)
        if num_mt:
            raise NotImplementedError("imitation-learning (mt) unrolls need the meta-gradient path "
                                      "(SURVEY.md 8f rank 2); use num_mt=0")
        self.num_mt = num_mt

    def meta_loss(self, make_loss, len_unroll, net_assignments=None, second_derivatives=False):
        graph = self._build_graph(make_loss, len_unroll, net_assignments, second_derivatives)
        return (self._handles(graph), graph.scale, graph.x, graph.constants, graph.subsets,
                [], [], [], [], [])

    def meta_minimize(self, make_loss, len_unroll, learning_rate=0.01, **kwargs):
        """DM/meta_dm_train.py:529-558: (MetaStep, scale, x, constants, subsets, loss_mt, steps_mt,
        update_mt, reset_mt, mt_labels, mt_inputs) -- the mt lists are empty (num_mt == 0)."""
        out = self.meta_loss(make_loss, len_unroll, **kwargs)
        self._graph.learning_rate = learning_rate
        step = _meta.MetaStep(_meta.Fetch(self._graph, "step"), *out[0][1:])
        return (step,) + tuple(out[1:5]) + ([], [], [], [], [], [])
