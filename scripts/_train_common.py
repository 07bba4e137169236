"""Meta-training driver shared by scripts/train_dm.py and scripts/train_rnnprop.py: the
reference's training harness (DM/train_dm.py:63-229, DM/train_rnnprop.py) re-hosted on
open_l2o_amd -- same flags, same schedule:

  * every epoch: reset, then num_steps // unroll_length truncated-BPTT segments, each one
    `sess.run([cost, update, step])` (DM/util.py:31-61);
  * every `evaluation_period` epochs: `evaluation_epochs` epochs without the meta-step; keep the
    best optimizer (`.l2l-{epoch}` / `.l2l-0`);
  * --if_cl: curriculum over horizons 100..3000 with save / restore of the best weights per stage
    (DM/train_dm.py:65-69, 198-226);
  * --if_scale: random per-coordinate rescaling of the optimizee (DM/util.py:40-54).
Imitation learning (--if_mt, DM/data_generator.py) is not implemented.
"""
import argparse
import os
import sys
from timeit import default_timer as timer

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
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--batch_size", type=int, default=None)
    p.add_argument("--num_dims", type=int, default=None)
    p.add_argument("--synthetic_mnist", type=int, default=0,
                   help="problems.mnist on N synthetic MNIST-shaped examples (no dataset ships offline)")
    if rnnprop:
        p.add_argument("--beta1", type=float, default=0.95)
        p.add_argument("--beta2", type=float, default=0.95)
    return p.parse_args()


class Trainer(object):
    def __init__(self, flags, rnnprop):
        self.f = flags
        self.rnnprop = rnnprop
        if flags.if_mt:
            raise NotImplementedError("--if_mt (imitation learning, DM/data_generator.py) is not implemented")
        if flags.seed:
            meta.set_random_seed(flags.seed)
        if flags.save_path and not os.path.exists(flags.save_path):
            os.mkdir(flags.save_path)
        opts = {k: v for k, v in (("batch_size", flags.batch_size), ("num_dims", flags.num_dims)) if v is not None}
        if getattr(flags, "synthetic_mnist", 0):
            from open_l2o_amd import problems
            opts["data"] = problems.synthetic_mnist(flags.synthetic_mnist)
        problem, net_config, assignments = util.get_config(flags.problem, net_name="RNNprop" if rnnprop else None,
                                                           problem_options=opts)
        kw = dict(learning_rate=flags.learning_rate, net_assignments=assignments,
                  second_derivatives=flags.second_derivatives)
        self.step_ph = None
        if rnnprop:
            self.optimizer = meta_rnnprop_train.MetaOptimizer(0, flags.beta1, flags.beta2, **net_config)
            out = self.optimizer.meta_minimize(problem, flags.unroll_length, **kw)
            self.minimize, self.scale, self.var_x, self.step_ph = out[0], out[1], out[2], out[5]
        else:
            self.optimizer = meta_dm_train.MetaOptimizer(0, **net_config)
            out = self.optimizer.meta_minimize(problem, flags.unroll_length, **kw)
            self.minimize, self.scale, self.var_x = out[0], out[1], out[2]

    def _epoch(self, sess, ops, n_unrolls, train):
        step, update, reset, cost_op, _ = self.minimize
        extra = dict(step=self.step_ph, unroll_len=self.f.unroll_length) if self.rnnprop else {}
        if train:
            extra.update(scale=self.scale, rd_scale=self.f.if_scale, rd_scale_bound=self.f.rd_scale_bound,
                         assign_func=lambda vals: [v.load(a) for v, a in zip(self.var_x, vals)], var_x=self.var_x)
        return util.run_epoch(sess, cost_op, ops, reset, n_unrolls, **extra)

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
        with MonitoredSession() as sess:
            sess.run(self.minimize.reset)
            for e in range(f.num_epochs):
                n_train = stages[stage] if f.if_cl else f.num_steps // f.unroll_length
                _, cost = self._epoch(sess, [update, step], n_train, train=True)
                print("training_loss={}".format(cost))
                if (e + 1) % f.evaluation_period:
                    continue
                n_eval += 1
                horizon = CURRICULUM[stage] if f.if_cl else f.num_steps
                eval_cost = self._evaluate(sess, stages[1:][stage] if f.if_cl else n_train)
                print("epoch={}, num_steps={}, eval_loss={}".format(e, horizon, eval_cost / f.evaluation_epochs),
                      flush=True)
                if not f.if_cl:
                    if eval_cost < best:
                        best = eval_cost
                        if f.save_path:
                            self.optimizer.save(sess, f.save_path, e + 1)
                            self.optimizer.save(sess, f.save_path, 0)
                            print("Saving optimizer of epoch {}...".format(e + 1))
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
        print("total time = {}s...".format(timer() - t0))


def main(rnnprop):
    Trainer(parse_flags(rnnprop), rnnprop).run()
