"""``tf.Session`` shim: evaluates the fetch handles ``MetaOptimizer.meta_loss`` returns.

    with Session() as sess:
        sess.run(reset)
        cost = sess.run([cost_op, update], feed_dict={step: 1})[0]

Semantics follow the reference's use of ``sess.run`` (DM/util.py:31-89,
DM/evaluate_dm.py:78-91): every ``run`` that fetches ``loss`` / ``fx`` / ``x`` /
``update`` executes ONE unroll from the current variables; ``update`` in the same
``run`` commits x_T / LSTM state (DM/meta.py:387-389); ``reset`` re-initialises the
variables (DM/meta.py:379-383).  Fetch structure (nested lists / tuples) is mirrored
in the result; ops evaluate to ``None``.
"""
from __future__ import annotations

from . import meta as _meta


def _flatten(fetches, out):
    if isinstance(fetches, (list, tuple)):
        for f in fetches:
            _flatten(f, out)
    else:
        out.append(fetches)
    return out


def _rebuild(fetches, values):
    if isinstance(fetches, (list, tuple)):
        res = [_rebuild(f, values) for f in fetches]
        return tuple(res) if isinstance(fetches, tuple) and not hasattr(fetches, "_fields") else (
            type(fetches)(*res) if hasattr(fetches, "_fields") else res)
    return values[id(fetches)]


class Session(object):
    """Minimal session: no graph, no devices -- it only sequences unrolls."""

    def __init__(self, config=None):
        self.config = config

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self):
        pass

    def run(self, fetches, feed_dict=None, _defer_loss=False):
        """_defer_loss (ours; util.run_epoch uses it for all but the last unroll of an epoch, whose cost is the only
        one DM/util.py:31-75 returns): a meta-training step is only ENQUEUED -- no host sync, the loss fetches come
        back as None -- so that the host runs ahead of the GPU instead of draining it once per unroll."""
        flat = _flatten(fetches, [])
        values = {}
        by_graph = {}
        for f in flat:
            if isinstance(f, _meta.Fetch):
                by_graph.setdefault(id(f.graph), (f.graph, []))[1].append(f)
            elif isinstance(f, _meta.Variable):
                values[id(f)] = f.eval()
            elif f is None:
                values[id(f)] = None
            else:
                raise TypeError("cannot fetch %r" % (f,))
        for graph, fs in by_graph.values():
            keys = [f.key for f in fs]
            do_reset = "reset" in keys
            needs_unroll = any(k in ("loss", "fx", "update", "fx_array", "step") or isinstance(k, tuple)
                               for k in keys)
            if do_reset:
                graph.reset()
            if "step" in keys:                              # the Adam meta-step (meta_minimize)
                res = graph.train_step(feed_dict, commit="update" in keys, learning_rate=graph.learning_rate,
                                       defer=_defer_loss)
            else:
                res = graph.execute(feed_dict, commit="update" in keys) if needs_unroll else {}
            for f in fs:
                if f.key in ("update", "reset", "step"):
                    values[id(f)] = None
                elif isinstance(f.key, tuple):
                    values[id(f)] = res["x"][f.key[1]]
                else:
                    values[id(f)] = res[f.key]
        return _rebuild(fetches, values)


# the reference opens ``ms.MonitoredSession()`` (DM/evaluate_dm.py:78); same object here
MonitoredSession = Session
