"""RNNProp training meta-optimizer -- the reference's ``DM/meta_rnnprop_train.py`` API
(forward unroll only; the imitation-learning "mt" losses and the Adam meta-step need the
meta-gradient).

``MetaOptimizer(num_mt, beta1, beta2, **net_config)``; ``meta_loss`` returns the
reference's tuple (DM/meta_rnnprop_train.py:592-593):
``(MetaLoss, scale, x, constants, subsets, seq_step, loss_mt, steps/update_mt, reset_mt,
mt_labels, mt_inputs, ...)`` with empty mt lists for ``num_mt == 0``.
"""
from . import meta as _meta
from .meta import MetaLoss, MetaStep, set_random_seed  # noqa: F401


class MetaOptimizer(_meta.MetaOptimizer):
    _rnnprop = True

    def __init__(self, num_mt, beta1, beta2, **kwargs):
        super(MetaOptimizer, self).__init__(**kwargs)
        if num_mt:
            raise NotImplementedError("imitation-learning (mt) unrolls need the meta-gradient path "
                                      "(SURVEY.md 8f rank 2); use num_mt=0")
        self.num_mt = num_mt
        self.beta1 = beta1
        self.beta2 = beta2

    def meta_loss(self, make_loss, len_unroll, net_assignments=None, second_derivatives=False):
        graph = self._build_graph(make_loss, len_unroll, net_assignments, second_derivatives)
        return (self._handles(graph), graph.scale, graph.x, graph.constants, graph.subsets, graph.step,
                [], [], [], [], [])

    def meta_minimize(self, make_loss, len_unroll, learning_rate=0.01, **kwargs):
        """DM/meta_rnnprop_train.py:595-624 (mt lists empty: num_mt == 0)."""
        out = self.meta_loss(make_loss, len_unroll, **kwargs)
        self._graph.learning_rate = learning_rate
        step = _meta.MetaStep(_meta.Fetch(self._graph, "step"), *out[0][1:])
        return (step,) + tuple(out[1:6]) + ([], [], [], [], [], [])
