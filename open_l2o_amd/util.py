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


def _rescale_iterate(sess, scale, bound, assign_func, var_x):
    """The random x-scaling augmentation of the train drivers (DM/util.py:40-54): draw
    r = exp(U[-bound, bound]) per coordinate, restart from x / r and feed r to the scale
    placeholders, so that the optimizee sees x * r == the sampled x0.  Returns the feed dict.

    The factors are drawn with the variables' GLOBAL shapes on rank 0 and broadcast
    (meta.synced_scale), a rank divides ITS shard of x by ITS slice of r."""
    from . import meta
    if scale is None or var_x is None or assign_func is None:
        raise ValueError("rd_scale=True needs scale, var_x and assign_func")
    factors = [meta.synced_scale(ph.shape, bound) for ph in scale]
    assign_func([sess.run(v) / meta.local_slice(v, r) for v, r in zip(var_x, factors)])
    return dict(zip(scale, factors))


def run_epoch(sess, cost_op, ops, reset, num_unrolls,
              scale=None, rd_scale=False, rd_scale_bound=3.0, assign_func=None, var_x=None,
              step=None, unroll_len=None,
              task_i=-1, data=None, label_pl=None, input_pl=None):
    """Runs one optimization epoch: `reset`, then `num_unrolls` x {cost_op + ops}; returns
    (seconds, cost of the last unroll).  Same signature and feeds as DM/util.py:31-75:
    task_i == -1 is a meta-training epoch on the optimizee (optionally with the random x-scaling
    of :40-54), task_i >= 0 an imitation epoch that feeds unroll i of a recorded teacher
    trajectory (`data`, :62-74); RNNProp's `step` placeholder gets i * unroll_len + 1 (:59-60)."""
    start = timer()
    sess.run(reset)
    if task_i == -1:
        base = _rescale_iterate(sess, scale, rd_scale_bound, assign_func, var_x) if rd_scale else {}

        def feed_of(i):
            return dict(base)
    else:
        if data is None or input_pl is None or label_pl is None:
            raise ValueError("an imitation epoch needs data, input_pl and label_pl")

        def feed_of(i):
            f = dict(zip(label_pl, data["labels"][i]))
            f.update(zip(input_pl, data["inputs"][i]))
            return f
    cost = None
    for i in range(num_unrolls):
        feed = feed_of(i)
        if step is not None:
            feed[step] = i * unroll_len + 1
        # only the LAST unroll's cost is returned (DM/util.py:75): the others are enqueued without a host sync
        if i + 1 < num_unrolls and task_i == -1 and getattr(sess, "run", None) is not None and _can_defer(sess):
            sess.run([cost_op] + ops, feed_dict=feed, _defer_loss=True)
        else:
            cost = sess.run([cost_op] + ops, feed_dict=feed)[0]
    return timer() - start, cost


def _can_defer(sess):
    import inspect
    try:
        return "_defer_loss" in inspect.signature(sess.run).parameters and not os.environ.get("L2O_NO_DEFER")
    except (TypeError, ValueError):
        return False


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
    elif problem_name == "mnist_deeper":                                 # DM/util.py:157-163: two hidden layers of 20
        if mode is None:
            mode = "train" if path is None else "test"
        problem = problems.mnist(**with_defaults(layers=(20, 20), mode=mode, activation="sigmoid"))
        net_config = {"cw": get_default_net_config(path)}
        net_assignments = None
    elif problem_name in ("mnist_conv", "cifar_conv", "lenet", "nas",
                          "vgg16", "cifar-multi", "confocal_microscopy_3d"):
        # neural-network / data-dependent optimizees of DM/util.py:144-230: the net config is
        # reproduced, the problem factory raises (out of the accelerated hot path).
        problem = getattr(problems, {"cifar_conv": "cifar10",
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
