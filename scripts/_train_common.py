"""Meta-training driver shared by scripts/train_dm.py and scripts/train_rnnprop.py: the
reference's training harness (DM/train_dm.py:63-229, DM/train_rnnprop.py) re-hosted on
open_l2o_amd -- same flags, same schedule:

  * every epoch: reset, then num_steps // unroll_length truncated-BPTT segments, each one
    `sess.run([cost, update, step])` (DM/util.py:31-61);
  * every `evaluation_period` epochs: `evaluation_epochs` epochs without the meta-step; keep the
    best optimizer (`.l2l-{epoch}` / `.l2l-0`);
  * --if_cl: curriculum over horizons 100..3000 with save / restore of the best weights per stage
    (DM/train_dm.py:65-69, 198-226);
  * --if_scale: random per-coordinate rescaling of the optimizee (DM/util.py:40-54);
  * --if_mt: with probability mt_ratio an epoch is an imitation epoch -- the trajectory of an
    analytic optimizer (--optimizers adam,rmsprop,nag; open_l2o_amd.data_generator) is recorded
    on a fresh problem and the networks regress its updates (DM/train_dm.py:112-146).
"""
import argparse
import os
import random
import sys
from timeit import default_timer as timer

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from open_l2o_amd import meta, meta_dm_train, meta_rnnprop_train, util  # noqa: E402
from open_l2o_amd.session import MonitoredSession  # noqa: E402

CURRICULUM = [100, 200, 500, 1000, 1500, 2000, 2500, 3000]       # DM/train_dm.py:66


def parse_flags(rnnprop):
    p = argparse.ArgumentParser()
    p.add_argument("--save_path", default=None, help="Path for saved meta-optimizer.")
    p.add_argument("--num_epochs", type=int, default=10000)
    p.add_argument("--evaluation_period", type=int, default=10 if rnnprop else 100)
    p.add_argument("--evaluation_epochs", type=int, default=20)
    p.add_argument("--num_steps", type=int, default=100, help="Number of optimization steps per epoch.")
    p.add_argument("--unroll_length", type=int, default=20, help="Meta-optimizer unroll length.")
    p.add_argument("--learning_rate", type=float, default=0.001)
    p.add_argument("--second_derivatives", action="store_true")
    p.add_argument("--problem", default="mnist")
    p.add_argument("--if_scale", action="store_true")
    p.add_argument("--rd_scale_bound", type=float, default=3.0)
    p.add_argument("--if_cl", action="store_true")
    p.add_argument("--min_num_eval", type=int, default=3)
    p.add_argument("--if_mt", action="store_true")
    p.add_argument("--num_mt", type=int, default=1)
    p.add_argument("--optimizers", default="adam", help="comma list of imitation teachers: adam, rmsprop, nag")
    p.add_argument("--mt_ratio", type=float, default=0.3)
    p.add_argument("--mt_ratios", default="0.3 0.3 0.3" if rnnprop else "0.0 0.1 0.3 0.3 0.3 0.3 0.3 0.3")
    p.add_argument("--k", type=int, default=1, help="teacher steps per recorded update (mt_k)")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--batch_size", type=int, default=None)
    p.add_argument("--num_dims", type=int, default=None)
    p.add_argument("--num_rows", type=int, default=None, help="lasso: rows of the sensing matrix (default: num_dims)")
    p.add_argument("--l", type=float, default=None, help="lasso: l1 weight (DM/problems.py:103 default 0.005)")
    p.add_argument("--max_seconds", type=float, default=None,
                   help="stop after the first evaluation past this wall time (bounded GPU leases)")
    p.add_argument("--synthetic_mnist", type=int, default=0,
                   help="problems.mnist on N synthetic MNIST-shaped examples (no dataset ships offline)")
    p.add_argument("--synthetic_label_noise", type=float, default=0.0,
                   help="fraction of the synthetic labels re-drawn uniformly (problems.synthetic_mnist)")
    p.add_argument("--synthetic_seed", type=int, default=0)
    if rnnprop:
        p.add_argument("--beta1", type=float, default=0.95)
        p.add_argument("--beta2", type=float, default=0.95)
    return p.parse_args()


class Trainer(object):
    def __init__(self, flags, rnnprop):
        self.f = flags
        self.rnnprop = rnnprop
        if flags.seed:
            meta.set_random_seed(flags.seed)
            # the harness' own draws (np.random in util.run_epoch's x-scale, random.random() for the
            # imitation ratio -- DM/util.py:40-54, DM/train_dm.py) follow the flag too: a seeded run repeats
            random.seed(flags.seed)
            np.random.seed(flags.seed)
        if flags.save_path and not os.path.exists(flags.save_path):
            os.mkdir(flags.save_path)
        opts = {k: v for k, v in (("batch_size", flags.batch_size), ("num_dims", flags.num_dims),
                                  ("num_rows", getattr(flags, "num_rows", None)), ("l", getattr(flags, "l", None)))
                if v is not None}
        if getattr(flags, "synthetic_mnist", 0):
            from open_l2o_amd import problems
            opts["data"] = problems.synthetic_mnist(flags.synthetic_mnist, seed=getattr(flags, "synthetic_seed", 0),
                                                    label_noise=getattr(flags, "synthetic_label_noise", 0.0))
        problem, net_config, assignments = util.get_config(flags.problem, net_name="RNNprop" if rnnprop else None,
                                                           problem_options=opts)
        kw = dict(learning_rate=flags.learning_rate, net_assignments=assignments,
                  second_derivatives=flags.second_derivatives)
        self.step_ph = None
        num_mt = getattr(flags, "num_mt", 1) if flags.if_mt else 0
        if rnnprop:
            self.optimizer = meta_rnnprop_train.MetaOptimizer(num_mt, flags.beta1, flags.beta2, **net_config)
            out = self.optimizer.meta_minimize(problem, flags.unroll_length, **kw)
            self.minimize, self.scale, self.var_x, self.step_ph = out[0], out[1], out[2], out[5]
            mt = out[6:]
        else:
            self.optimizer = meta_dm_train.MetaOptimizer(num_mt, **net_config)
            out = self.optimizer.meta_minimize(problem, flags.unroll_length, **kw)
            self.minimize, self.scale, self.var_x = out[0], out[1], out[2]
            mt = out[5:]
        self.loss_mt, self.steps_mt, self.update_mt, self.reset_mt, self.mt_labels, self.mt_inputs = mt
        self.data_mt = None
        if flags.if_mt:                                        # DM/train_dm.py:91-96
            from open_l2o_amd.data_generator import data_loader
            self.data_mt = data_loader(problem, self.var_x, out[3], out[4], self.scale, getattr(flags, "optimizers", "adam"),
                                       flags.unroll_length)
            if len(self.data_mt.optimizers) < num_mt:
                raise ValueError("--num_mt %d needs as many --optimizers" % num_mt)

    def _epoch(self, sess, ops, n_unrolls, train):
        step, update, reset, cost_op, _ = self.minimize
        extra = dict(step=self.step_ph, unroll_len=self.f.unroll_length) if self.rnnprop else {}
        if train:
            extra.update(scale=self.scale, rd_scale=self.f.if_scale, rd_scale_bound=self.f.rd_scale_bound,
                         assign_func=lambda vals: [v.load(a) for v, a in zip(self.var_x, vals)], var_x=self.var_x)
        return util.run_epoch(sess, cost_op, ops, reset, n_unrolls, **extra)

    def _assign(self, vals):
        for v, a in zip(self.var_x, vals):
            v.load(a)

    def _imitation_epoch(self, sess, task_i, n_unrolls):
        """DM/train_dm.py:134-146."""
        f = self.f
        data = self.data_mt.get_data(task_i, sess, n_unrolls, self._assign, f.rd_scale_bound, if_scale=f.if_scale,
                                     mt_k=getattr(f, "k", 1))
        extra = dict(step=self.step_ph, unroll_len=f.unroll_length) if self.rnnprop else {}
        return util.run_epoch(sess, self.loss_mt[task_i], [self.update_mt[task_i], self.steps_mt[task_i]],
                              self.reset_mt[task_i], n_unrolls, task_i=task_i, data=data,
                              label_pl=self.mt_labels[task_i], input_pl=self.mt_inputs[task_i], **extra)

    def _evaluate(self, sess, n_unrolls):
        update = self.minimize.update
        return sum(self._epoch(sess, [update], n_unrolls, train=False)[1] for _ in range(self.f.evaluation_epochs))

    def run(self):
        f = self.f
        stages = [n // f.unroll_length for n in CURRICULUM] if f.if_cl else None
        stage = 0
        step, update = self.minimize.step, self.minimize.update
        best, n_eval, improved = float("inf"), 0, False
        t0 = timer()
        mt_ratios = [float(r) for r in getattr(f, "mt_ratios", "0.3").split()]
        mti = -1
        stop = False
        with MonitoredSession() as sess:
            for rst in [self.minimize.reset] + self.reset_mt:
                sess.run(rst)
            for e in range(f.num_epochs):
                n_train = stages[stage] if f.if_cl else f.num_steps // f.unroll_length
                task_i = -1
                if f.if_mt:                                    # pick a task, DM/train_dm.py:112-127
                    mt_ratio = (mt_ratios[min(stage, len(mt_ratios) - 1)] if f.if_cl else getattr(f, "mt_ratio", 0.3))
                    if random.random() < mt_ratio:
                        mti = (mti + 1) % f.num_mt
                        task_i = mti
                if task_i == -1:
                    _, cost = self._epoch(sess, [update, step], n_train, train=True)
                else:
                    _, cost = self._imitation_epoch(sess, task_i, n_train)
                print("training_loss={}".format(cost))
                if (e + 1) % f.evaluation_period:
                    continue
                n_eval += 1
                horizon = CURRICULUM[stage] if f.if_cl else f.num_steps
                eval_cost = self._evaluate(sess, stages[1:][stage] if f.if_cl else n_train)
                print("epoch={}, num_steps={}, eval_loss={}".format(e, horizon, eval_cost / f.evaluation_epochs),
                      flush=True)
                if getattr(f, "max_seconds", None) and timer() - t0 > f.max_seconds:
                    stop = True
                if not f.if_cl:
                    if eval_cost < best:
                        best = eval_cost
                        if f.save_path:
                            self.optimizer.save(sess, f.save_path, e + 1)
                            self.optimizer.save(sess, f.save_path, 0)
                            print("Saving optimizer of epoch {}...".format(e + 1))
                    if stop:
                        break
                    continue
                # curriculum: advance when a stage stopped improving (DM/train_dm.py:198-226)
                if eval_cost < best:
                    best, improved = eval_cost, True
                    if f.save_path:
                        self.optimizer.save(sess, f.save_path, stage)
                        self.optimizer.save(sess, f.save_path, 0)
                elif n_eval >= f.min_num_eval and improved:
                    if f.save_path:
                        self.optimizer.restore(sess, f.save_path, stage)
                    n_eval, improved = 0, False
                    stage = stage + 1 if stage + 1 < len(stages) - 1 else len(stages) - 2
                    best = self._evaluate(sess, stages[1:][stage])
                    print("epoch={}, num_steps={}, eval loss={}".format(e, CURRICULUM[stage],
                                                                       best / f.evaluation_epochs), flush=True)
                elif n_eval >= f.min_num_eval and not improved:
                    print("no improve during curriculum {} --> stop".format(stage))
                    break
                if stop:
                    break
        print("total time = {}s...".format(timer() - t0))


def main(rnnprop):
    Trainer(parse_flags(rnnprop), rnnprop).run()
