"""Learning 2 Learn problems -- the optimizee registry of the reference
(``DM/problems.py``, DM = /root/reference/Model_Free_L2O/"L2O-DM and L2O-RNNProp"/),
session-less.

Same module-level factories, same argument names and defaults.  Each factory
returns a zero-argument ``build`` object, exactly like the reference returns a
``build`` closure; calling it declares the problem's variables through
:func:`get_variable` (the analogue of ``tf.get_variable``, which is what
``meta._get_variables`` intercepts at DM/meta.py:102-128) and returns a
:class:`Loss` description.  The arithmetic itself -- forward value and gradient
(DM/meta.py:322, 344) -- is done by the HIP kernels selected through ``Loss.terms``
(``l2o_problem_fg`` / ``l2o_unroll``); nothing here computes on the CPU.

Extensions over the reference (all optional, defaults unchanged):
  * ``lasso(..., num_rows=None)``  rectangular per-problem A like ``lasso_fixed``;
  * ``quadratic/lasso/rastrigin(..., data=dict)``  inject W/y/x0 arrays (parity tests).
"""
from __future__ import annotations

import collections
import sys

import numpy as np

from . import _abi

# ---------------------------------------------------------------------------
# variable declaration plumbing (tf.get_variable analogue)
# ---------------------------------------------------------------------------
VarDecl = collections.namedtuple("VarDecl", "name shape initializer trainable")
# one analytic loss term: kind (an _abi.PROB_* id), the trainable variable it is a
# function of, its constants by role, hyper-parameters and an ensemble weight
Term = collections.namedtuple("Term", "kind var consts hyper weight")
Loss = collections.namedtuple("Loss", "variables terms")

_scope = []          # variable_scope stack (ensemble uses "problem_i")
_decls = None        # active collection list while a build() runs


def _full_name(name):
    return "/".join(_scope + [name])


class _SharedVarDecl(VarDecl):
    """A constant with ONE copy for the whole batch (leading dimension 1; never batch-sharded)."""
    __slots__ = ()
    shared = True


def get_variable(name, shape, dtype="float32", initializer=None, trainable=True, shared=False):
    """Declare an optimizee variable (the reference calls ``tf.get_variable`` here;
    DM/meta.py:88-155 relies on that to harvest / substitute variables)."""
    if dtype not in ("float32", np.float32):
        raise ValueError("only float32 optimizees are implemented (got %r)" % (dtype,))
    cls = _SharedVarDecl if shared else VarDecl
    decl = cls(_full_name(name), tuple(int(s) for s in shape), initializer, bool(trainable))
    if _decls is not None:
        _decls.append(decl)
    return decl


# initializer descriptors (resolved on the device by meta.Variable)
def random_normal_initializer(mean=0.0, stddev=1.0):
    return ("normal", float(mean), float(stddev))


def random_uniform_initializer(minval=0.0, maxval=1.0):
    return ("uniform", float(minval), float(maxval))


def ones_initializer():
    return ("ones",)


def zeros_initializer():
    return ("zeros",)


def constant_initializer(value):
    return ("constant", np.asarray(value, dtype=np.float32))


class _Build(object):
    """Callable returned by the factories (the reference returns the closure ``build``)."""

    def __init__(self, name, fn):
        self.__name__ = name
        self._fn = fn

    def __call__(self):
        global _decls
        outer = _decls
        _decls = []
        try:
            terms = self._fn()
            variables = _decls
        finally:
            _decls = outer
        if outer is not None:          # nested (ensemble): hand declarations to the parent
            outer.extend(variables)
        return Loss(variables, terms)


def _maybe_const(data, key, shape, default):
    if data is not None and key in data:
        arr = np.asarray(data[key], dtype=np.float32).reshape(shape)
        return constant_initializer(arr)
    return default


# ---------------------------------------------------------------------------
# the registry (names, arguments and defaults of DM/problems.py)
# ---------------------------------------------------------------------------
def simple():
    """Simple problem: f(x) = x^2.  DM/problems.py:41-53."""

    def build():
        x = get_variable("x", shape=[], initializer=ones_initializer())
        return [Term(_abi.PROB_SIMPLE, x, {}, {}, 1.0)]

    return _Build("simple", build)


def simple_multi_optimizer(num_dims=2):
    """Multidimensional simple problem.  DM/problems.py:56-70."""

    def build():
        coords = [get_variable("x_{}".format(i), shape=[], initializer=ones_initializer())
                  for i in range(num_dims)]
        return [Term(_abi.PROB_SIMPLE, c, {}, {}, 1.0) for c in coords]

    return _Build("simple_multi_optimizer", build)


def quadratic(batch_size=128, num_dims=10, stddev=0.01, dtype="float32", data=None):
    """Quadratic problem: f(x) = ||Wx - y||.  DM/problems.py:73-101."""

    def build():
        x = get_variable("x", shape=[batch_size, num_dims], dtype=dtype,
                         initializer=_maybe_const(data, "x", [batch_size, num_dims],
                                                  random_normal_initializer(stddev=stddev)))
        w = get_variable("w", shape=[batch_size, num_dims, num_dims], dtype=dtype,
                         initializer=_maybe_const(data, "w", [batch_size, num_dims, num_dims],
                                                  random_uniform_initializer()), trainable=False)
        y = get_variable("y", shape=[batch_size, num_dims], dtype=dtype,
                         initializer=_maybe_const(data, "y", [batch_size, num_dims],
                                                  random_uniform_initializer()), trainable=False)
        return [Term(_abi.PROB_QUADRATIC, x, {"W": w, "y": y}, {}, 1.0)]

    return _Build("quadratic", build)


def lasso(batch_size=128, num_dims=10, stddev=0.01, l=0.005, dtype="float32", num_rows=None, data=None):
    """lasso problem: f(x) = 0.5*||Wx - y||2 + lamada *||x||1.  DM/problems.py:103-134."""
    rows = num_dims if num_rows is None else int(num_rows)

    def build():
        x = get_variable("x", shape=[batch_size, num_dims], dtype=dtype,
                         initializer=_maybe_const(data, "x", [batch_size, num_dims],
                                                  random_normal_initializer(stddev=stddev)))
        w = get_variable("w", shape=[batch_size, rows, num_dims], dtype=dtype,
                         initializer=_maybe_const(data, "w", [batch_size, rows, num_dims],
                                                  random_uniform_initializer()), trainable=False)
        y = get_variable("y", shape=[batch_size, rows, 1], dtype=dtype,
                         initializer=_maybe_const(data, "y", [batch_size, rows, 1],
                                                  random_uniform_initializer()), trainable=False)
        return [Term(_abi.PROB_LASSO, x, {"W": w, "y": y}, {"l1": float(l)}, 1.0)]

    return _Build("lasso", build)


def lasso_fixed(data_A, data_b, stddev=0.01, l=0.005, dtype="float32"):
    """lasso problem on given data A [B,M,N], b [B,M,1].  DM/problems.py:137-175.
    A 2-D ``data_A`` [M,N] is ONE sensing matrix shared by all B problems (the
    kernels then read it with batch stride 0 -- 512 KiB instead of 128 MiB for config 3)."""
    a = np.asarray(data_A, dtype=np.float32)
    b = np.asarray(data_b, dtype=np.float32)
    shared = a.ndim == 2
    if shared:
        a = a[None]
    if a.ndim != 3 or b.ndim != 3 or b.shape[1] != a.shape[1] or (not shared and b.shape[0] != a.shape[0]):
        raise ValueError("lasso_fixed expects data_A [B,M,N] (or one shared [M,N]) and data_b [B,M,1]")

    def build():
        x = get_variable("x", shape=[b.shape[0], a.shape[2]], dtype=dtype,
                         initializer=random_normal_initializer(stddev=stddev))
        w = get_variable("w", shape=a.shape, dtype=dtype, initializer=constant_initializer(a),
                         trainable=False, shared=shared)
        y = get_variable("y", shape=b.shape, dtype=dtype, initializer=constant_initializer(b),
                         trainable=False)
        return [Term(_abi.PROB_LASSO, x, {"W": w, "y": y}, {"l1": float(l)}, 1.0)]

    return _Build("lasso_fixed", build)


def rastrigin(batch_size=128, num_dims=10, alpha=10, stddev=1, dtype="float32", data=None):
    """Rastrigin-like problem.  DM/problems.py:177-213."""

    def build():
        x = get_variable("x", shape=[batch_size, num_dims, 1], dtype=dtype,
                         initializer=_maybe_const(data, "x", [batch_size, num_dims, 1],
                                                  random_normal_initializer(stddev=stddev)))
        A = get_variable("A", shape=[batch_size, num_dims, num_dims], dtype=dtype,
                         initializer=_maybe_const(data, "A", [batch_size, num_dims, num_dims],
                                                  random_normal_initializer(stddev=stddev)), trainable=False)
        B = get_variable("B", shape=[batch_size, num_dims, 1], dtype=dtype,
                         initializer=_maybe_const(data, "B", [batch_size, num_dims, 1],
                                                  random_normal_initializer(stddev=stddev)), trainable=False)
        C = get_variable("C", shape=[batch_size, num_dims, 1], dtype=dtype,
                         initializer=_maybe_const(data, "C", [batch_size, num_dims, 1],
                                                  random_normal_initializer(stddev=stddev)), trainable=False)
        return [Term(_abi.PROB_RASTRIGIN, x, {"W": A, "y": B, "C": C}, {"alpha": float(alpha)}, 1.0)]

    return _Build("rastrigin", build)


def square_cos(batch_size=128, num_dims=10, stddev=0.01, dtype="float32", data=None):
    """f = mean_b [ ||w x - y||^2 - sum_i (wcos (10 cos(2*3.1415926 x)))_i + 10 D ].  DM/problems.py:959-994."""

    def build():
        x = get_variable("x", shape=[batch_size, num_dims], dtype=dtype,
                         initializer=_maybe_const(data, "x", [batch_size, num_dims],
                                                  random_normal_initializer(stddev=stddev)))
        w = get_variable("w", shape=[batch_size, num_dims, num_dims], dtype=dtype,
                         initializer=_maybe_const(data, "w", [batch_size, num_dims, num_dims],
                                                  random_uniform_initializer()), trainable=False)
        y = get_variable("y", shape=[batch_size, num_dims], dtype=dtype,
                         initializer=_maybe_const(data, "y", [batch_size, num_dims],
                                                  random_uniform_initializer()), trainable=False)
        wcos = get_variable("wcos", shape=[batch_size, num_dims, num_dims], dtype=dtype,
                            initializer=_maybe_const(data, "wcos", [batch_size, num_dims, num_dims],
                                                     random_uniform_initializer()), trainable=False)
        return [Term(_abi.PROB_SQUARE_COS, x, {"W": w, "y": y, "wcos": wcos}, {}, 1.0)]

    return _Build("square_cos", build)


def ensemble(problems, weights=None):
    """Ensemble of problems: sum of (weighted) losses.  DM/problems.py:215-245."""
    if weights and len(weights) != len(problems):
        raise ValueError("len(weights) != len(problems)")
    build_fns = [getattr(sys.modules[__name__], p["name"])(**p["options"]) for p in problems]

    def build():
        terms = []
        for i, build_fn in enumerate(build_fns):
            _scope.append("problem_{}".format(i))
            try:
                sub = build_fn()
            finally:
                _scope.pop()
            for t in sub.terms:
                terms.append(t._replace(weight=t.weight * (weights[i] if weights else 1.0)))
        return terms

    return _Build("ensemble", build)


_nn_initializers = {                                # DM/problems.py:35-38
    "w": random_normal_initializer(mean=0, stddev=0.01),
    "b": random_normal_initializer(mean=0, stddev=0.01),
}


def synthetic_mnist(num_examples=2048, seed=0, label_noise=0.0):
    """A deterministic stand-in for the MNIST arrays (no dataset ships with this repo and
    there is no network): images [N,28,28,1] in [0,1], labels [N] in 0..9.
    ``label_noise``: that fraction of the labels is re-drawn uniformly AFTER the images were formed.  The clean
    set (0.0) is separable -- a trained optimizer drives the MLP's cross-entropy from ln 10 to ~1e-5 in 200 steps,
    where a relative error of the loss measures nothing; with 0.1 the achievable loss is ~0.5 (about where the
    784-20-10 MLP gets on the real digits in 200 steps), which is what bench.py's config 5 and the
    trained-parity test use."""
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, 10, size=num_examples).astype(np.int64)
    protos = rng.random((10, 28 * 28)) < 0.2
    images = (protos[labels] * rng.random((num_examples, 28 * 28))).astype(np.float32)
    if label_noise > 0.0:
        flip = rng.random(num_examples) < label_noise
        labels = np.where(flip, rng.integers(0, 10, size=num_examples), labels).astype(np.int64)
    return {"images": images.reshape(num_examples, 28, 28, 1), "labels": labels}


def _load_mnist(mode):
    import os
    path = os.environ.get("L2O_MNIST_NPZ")
    if not path:
        raise FileNotFoundError(
            "problems.mnist needs the MNIST arrays: pass data={'images': [N,28,28,1], 'labels': [N]} "
            "(problems.synthetic_mnist() gives an offline stand-in) or point L2O_MNIST_NPZ at an .npz with "
            "'{mode}_images' / '{mode}_labels' (the reference downloads them, DM/problems.py:267-272)")
    z = np.load(path)
    return {"images": z["%s_images" % mode], "labels": z["%s_labels" % mode]}


def mnist(layers, activation="sigmoid", batch_size=128, mode="train", data=None, sampler=None):
    """Mnist classification with a multi-layer perceptron.  DM/problems.py:254-288.

    ``layers=(20,)`` (util.get_config("mnist")) has the fused persistent unrolls; more hidden layers (``(20, 20)``:
    "mnist_deeper", DM/util.py:157-163; up to three of <= 32 units) run on the step-granular kernels (l2o_mlp_deep_fg).
    ``data`` / ``sampler(n_evals, batch, n_data) -> indices`` are ours (offline / parity tests);
    by default every evaluation draws a fresh uniform minibatch like the reference (:282-284)."""
    if activation not in ("sigmoid", "relu"):
        raise ValueError("{} activation not supported".format(activation))
    layers = tuple(layers)
    if not 1 <= len(layers) <= 3 or any(not 1 <= int(h) <= 32 for h in layers):
        raise NotImplementedError("problems.mnist is implemented for one to three hidden layers of at most 32 units "
                                  "(got layers=%r)" % (layers,))
    if data is None:
        data = _load_mnist(mode)
    images = np.asarray(data["images"], np.float32)
    labels = np.asarray(data["labels"]).astype(np.int32)
    n_in = int(np.prod(images.shape[1:]))

    def build():
        _scope.append("mlp")
        try:
            widths = [n_in] + [int(h) for h in layers] + [10]       # snt.nets.MLP(list(layers) + [10]), DM/problems.py:275
            vs = []
            for l in range(len(widths) - 1):
                vs.append(get_variable("linear_%d/w" % l, [widths[l], widths[l + 1]], initializer=_nn_initializers["w"]))
                vs.append(get_variable("linear_%d/b" % l, [widths[l + 1]], initializer=_nn_initializers["b"]))
        finally:
            _scope.pop()
        hyper = {"images": images, "labels": labels, "batch_size": int(batch_size), "activation": activation,
                 "sampler": sampler, "layers": tuple(int(h) for h in layers)}
        return [Term(_abi.PROB_MLP, tuple(vs), {}, hyper, 1.0)]

    return _Build("mnist", build)


def _not_on_hot_path(name, where):
    def factory(*args, **kwargs):
        raise NotImplementedError(
            "problems.%s (%s) is outside the accelerated hot path of this build; see DESIGN.md "
            "'out of scope'" % (name, where))
    factory.__name__ = name
    return factory


# neural-network / data-dependent optimizees of the reference (conv nets, TF queues,
# downloads).  Declared so that `getattr(problems, name)` fails with a clear message.
mnist_conv = _not_on_hot_path("mnist_conv", "DM/problems.py:291")
cifar10 = _not_on_hot_path("cifar10", "DM/problems.py:369")
LeNet = _not_on_hot_path("LeNet", "DM/problems.py:461")
NAS = _not_on_hot_path("NAS", "DM/problems.py:540")
vgg16_cifar10 = _not_on_hot_path("vgg16_cifar10", "DM/problems.py:637")
confocal_microscopy_3d = _not_on_hot_path("confocal_microscopy_3d", "DM/problems.py:701-956")
