"""Learning 2 Learn utils -- the reference's ``DM/util.py`` harness
(DM = /root/reference/Model_Free_L2O/"L2O-DM and L2O-RNNProp"/): ``run_epoch``,
``run_eval_epoch``, ``print_stats``, ``get_default_net_config``, ``get_config`` with the
same names, signatures and problem / net registry.  ``get_config`` additionally accepts
``problem_options`` (dict) to override the hard-coded problem sizes
(the reference fixes e.g. quadratic to batch 128 x 10 dims, DM/util.py:137).
"""
from __future__ import absolute_import, division, print_function

from timeit import default_timer as timer

import os
import numpy as np

from . import problems


def run_epoch(sess, cost_op, ops, reset, num_unrolls,
              scale=None, rd_scale=False, rd_scale_bound=3.0, assign_func=None, var_x=None,
              step=None, unroll_len=None,
              task_i=-1, data=None, label_pl=None, input_pl=None):
    """Runs one optimization epoch.  DM/util.py:31-75."""
    start = timer()
    sess.run(reset)
    cost = None
    if task_i == -1:
        if rd_scale:
            assert scale is not None
            r_scale = []
            for k in scale:
                r_scale.append(np.exp(np.random.uniform(-rd_scale_bound, rd_scale_bound, size=k.shape)))
            assert var_x is not None
            k_value_list = []
            for k_id in range(len(var_x)):
                k_value = sess.run(var_x[k_id])
                k_value = k_value / r_scale[k_id]
                k_value_list.append(k_value)
            assert assign_func is not None
            assign_func(k_value_list)
            feed_rs = {p: v for p, v in zip(scale, r_scale)}
        else:
            feed_rs = {}
        feed_dict = feed_rs
        for i in range(num_unrolls):
            if step is not None:
                feed_dict[step] = i * unroll_len + 1
            cost = sess.run([cost_op] + ops, feed_dict=feed_dict)[0]
    else:                                               # imitation epoch, DM/util.py:62-74
        assert data is not None
        assert input_pl is not None
        assert label_pl is not None
        feed_dict = {}
        for ri in range(num_unrolls):
            for pl, dat in zip(label_pl, data["labels"][ri]):
                feed_dict[pl] = dat
            for pl, dat in zip(input_pl, data["inputs"][ri]):
                feed_dict[pl] = dat
            if step is not None:
                feed_dict[step] = ri * unroll_len + 1
            cost = sess.run([cost_op] + ops, feed_dict=feed_dict)[0]
    return timer() - start, cost


def run_eval_epoch(sess, cost_op, ops, num_unrolls, step=None, unroll_len=None):
    """Runs one optimization epoch.  DM/util.py:78-89."""
    start = timer()
    total_cost = []
    feed_dict = {}
    # The whole epoch as ONE unroll when that is exactly the same computation (deterministic optimizee,
    # cost_op = the unroll's final loss, ops = its update): `num_unrolls` host round trips of one or a few
    # steps each are what this path spends its time on in the reference.
    graph = getattr(cost_op, "graph", None)
    flat = []

    def _flat(o):
        for e in o:
            _flat(e) if isinstance(e, (list, tuple)) else flat.append(e)
    _flat(ops)                                              # (MetaLoss.update is a list of ops, like the reference's)
    if (num_unrolls > 1 and not os.environ.get("L2O_EVAL_STEPWISE") and getattr(cost_op, "key", None) == "fx"
            and len(flat) == 1 and getattr(flat[0], "key", None) == "update" and getattr(flat[0], "graph", None) is graph
            and hasattr(graph, "execute_many") and graph.many_ok()
            and (step is None or unroll_len == graph.len_unroll)):
        total_cost = graph.execute_many(num_unrolls)
        return timer() - start, total_cost
    for i in range(num_unrolls):
        if step is not None:
            feed_dict[step] = i * unroll_len + 1
        cost = sess.run([cost_op] + ops, feed_dict=feed_dict)[0]
        total_cost.append(cost)
    return timer() - start, total_cost


def print_stats(header, total_error, total_time, n):
    """Prints experiment statistics.  DM/util.py:92-96."""
    print(header)
    print("Log Mean Final Error: {:.2f}".format(np.log10(total_error / n)))
    print("Mean epoch time: {:.2f} s".format(total_time / n))


def get_default_net_config(path):
    """DM/util.py:99-109."""
    return {
        "net": "CoordinateWiseDeepLSTM",
        "net_options": {
            "layers": (20, 20),
            "preprocess_name": "LogAndSign",
            "preprocess_options": {"k": 5},
            "scale": 0.01,
        },
        "net_path": path
    }


def _cw2020(path):
    return {"cw": {
        "net": "CoordinateWiseDeepLSTM",
        "net_options": {"layers": (20, 20)},
        "net_path": path
    }}


def get_config(problem_name, path=None, mode=None, num_hidden_layer=None, net_name=None,
               problem_options=None):
    """Returns problem configuration: (problem, net_config, net_assignments).  DM/util.py:112-265."""
    opts = dict(problem_options or {})

    def with_defaults(**defaults):
        defaults.update(opts)
        return defaults

    if problem_name == "simple":
        problem = problems.simple()
        net_config = {"cw": {
            "net": "CoordinateWiseDeepLSTM",
            "net_options": {"layers": (), "initializer": "zeros"},
            "net_path": path
        }}
        net_assignments = None
    elif problem_name == "simple-multi":
        problem = problems.simple_multi_optimizer()
        net_config = {
            "cw": {
                "net": "CoordinateWiseDeepLSTM",
                "net_options": {"layers": (), "initializer": "zeros"},
                "net_path": path
            },
            "adam": {
                "net": "Adam",
                "net_options": {"learning_rate": 0.01}
            }
        }
        net_assignments = [("cw", ["x_0"]), ("adam", ["x_1"])]
    elif problem_name == "quadratic":
        problem = problems.quadratic(**with_defaults(batch_size=128, num_dims=10))
        net_config = _cw2020(path)
        net_assignments = None
    elif problem_name == "square_cos":
        problem = problems.square_cos(**with_defaults(batch_size=128, num_dims=2))
        net_config = _cw2020(path)
        net_assignments = None
    elif problem_name == "rastrigin":
        problem = problems.rastrigin(**with_defaults(batch_size=128, num_dims=2))
        net_config = _cw2020(path)
        net_assignments = None
    elif problem_name == "lasso":
        problem = problems.lasso(**with_defaults(batch_size=128, num_dims=2))
        net_config = _cw2020(path)
        net_assignments = None
    elif problem_name in ("mnist", "mnist_relu"):                      # DM/util.py:144-155
        if mode is None:
            mode = "train" if path is None else "test"
        problem = problems.mnist(**with_defaults(layers=(20,), mode=mode,
                                                 activation="sigmoid" if problem_name == "mnist" else "relu"))
        net_config = {"cw": get_default_net_config(path)}
        net_assignments = None
    elif problem_name in ("mnist_deeper", "mnist_conv", "cifar_conv", "lenet", "nas",
                          "vgg16", "cifar-multi", "confocal_microscopy_3d"):
        # neural-network / data-dependent optimizees of DM/util.py:144-230: the net config is
        # reproduced, the problem factory raises (out of the accelerated hot path).
        if problem_name == "mnist_deeper":
            problems.mnist(layers=(20, 20), data=problems.synthetic_mnist(8))     # raises: one hidden layer only
        problem = getattr(problems, {"mnist_deeper": "mnist", "cifar_conv": "cifar10",
                                     "lenet": "LeNet", "nas": "NAS", "vgg16": "vgg16_cifar10",
                                     "cifar-multi": "cifar10"}.get(problem_name, problem_name))()
        net_config = {"cw": get_default_net_config(path)}
        net_assignments = None
    else:
        raise ValueError("{} is not a valid problem".format(problem_name))

    if net_name == "RNNprop":
        default_config = {
            "net": "RNNprop",
            "net_options": {
                "layers": (20, 20),
                "preprocess_name": "fc",
                "preprocess_options": {"dim": 20},
                "scale": 0.01,
                "tanh_output": True
            },
            "net_path": path
        }
        net_config = {"rp": default_config}

    return problem, net_config, net_assignments
