"""RNNProp training meta-optimizer -- the reference's ``DM/meta_rnnprop_train.py`` API: forward
unroll with x-scale and ``step`` placeholders, the Adam meta-step and ``num_mt`` imitation-learning
("mt") unrolls with their own moments (DM/meta_rnnprop_train.py:437-555).

``MetaOptimizer(num_mt, beta1, beta2, **net_config)``; ``meta_loss`` returns the reference's
tuple (DM/meta_rnnprop_train.py:592-593): ``(MetaLoss, scale, x, constants, subsets, seq_step,
loss_mt, update_mt, reset_mt, mt_labels, mt_inputs)``.
"""
from . import meta as _meta
from .meta import MetaLoss, MetaStep, set_random_seed  # noqa: F401


class MetaOptimizer(_meta.MetaOptimizer):
    _rnnprop = True

    def __init__(self, num_mt, beta1, beta2, **kwargs):
        super(MetaOptimizer, self).__init__(**kwargs)
        self.num_mt = int(num_mt)
        self.beta1 = beta1
        self.beta2 = beta2

    def meta_loss(self, make_loss, len_unroll, net_assignments=None, second_derivatives=False):
        graph = self._build_graph(make_loss, len_unroll, net_assignments, second_derivatives)
        loss_mt, _, update_mt, reset_mt, mt_labels, mt_inputs = _meta.make_mt_handles(graph, self.num_mt)
        return (self._handles(graph), graph.scale, graph.x, graph.constants, graph.subsets, graph.step,
                loss_mt, update_mt, reset_mt, mt_labels, mt_inputs)

    def meta_minimize(self, make_loss, len_unroll, learning_rate=0.01, **kwargs):
        """DM/meta_rnnprop_train.py:595-624."""
        out = self.meta_loss(make_loss, len_unroll, **kwargs)
        graph = self._graph
        graph.learning_rate = learning_rate
        loss_mt, steps_mt, update_mt, reset_mt, mt_labels, mt_inputs = _meta.make_mt_handles(graph, self.num_mt)
        for m in graph.mt:
            m.learning_rate = learning_rate
        step = _meta.MetaStep(_meta.Fetch(graph, "step"), *out[0][1:])
        return (step,) + tuple(out[1:6]) + (loss_mt, steps_mt, update_mt, reset_mt, mt_labels, mt_inputs)
