#!/usr/bin/env python
"""Learning 2 Learn evaluation (RNNProp) -- the reference's DM/evaluate_rnnprop.py re-hosted
on open_l2o_amd (same flags incl. --beta1/--beta2, same flow; `step` is fed by
util.run_eval_epoch exactly like DM/util.py:84-87)."""
import argparse
import logging
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from open_l2o_amd import meta_rnnprop_eval as meta, util  # noqa: E402
from open_l2o_amd.session import MonitoredSession  # noqa: E402


def main():
    flags = argparse.ArgumentParser()
    flags.add_argument("--optimizer", default="L2L")
    flags.add_argument("--problem", default="simple")
    flags.add_argument("--path", default=None)
    flags.add_argument("--output_path", default=None)
    flags.add_argument("--num_epochs", type=int, default=1)
    flags.add_argument("--num_steps", type=int, default=10000)
    flags.add_argument("--learning_rate", type=float, default=0.001)
    flags.add_argument("--seed", type=int, default=None)
    flags.add_argument("--beta1", type=float, default=0.95)
    flags.add_argument("--beta2", type=float, default=0.95)
    flags.add_argument("--unroll_len", type=int, default=1)
    flags.add_argument("--batch_size", type=int, default=None)
    flags.add_argument("--num_dims", type=int, default=None)
    flags.add_argument("--synthetic_mnist", type=int, default=0,
                       help="problems.mnist on N synthetic MNIST-shaped examples (no dataset ships offline)")
    flags.add_argument("--replicas", type=int, default=1,
                       help="(ours) evaluate this many independent instances of the optimizee TOGETHER (own initial weights, "
                            "own minibatches; problems.mnist on the MI355X: eight per launch, one per XCD -- "
                            "open_l2o_amd.replicas.Replicas); the loss record holds one list per instance")
    FLAGS = flags.parse_args()

    num_unrolls = FLAGS.num_steps // FLAGS.unroll_len
    if FLAGS.seed:
        meta.set_random_seed(FLAGS.seed)
    opts = {k: v for k, v in (("batch_size", FLAGS.batch_size), ("num_dims", FLAGS.num_dims)) if v is not None}
    if FLAGS.synthetic_mnist:
        from open_l2o_amd import problems
        opts["data"] = problems.synthetic_mnist(FLAGS.synthetic_mnist)
    problem, net_config, net_assignments = util.get_config(FLAGS.problem, FLAGS.path, net_name="RNNprop",
                                                           problem_options=opts)
    if FLAGS.optimizer != "L2L":
        raise ValueError("{} is not a valid optimizer".format(FLAGS.optimizer))
    if FLAGS.path is None:
        logging.warning("Evaluating untrained L2L optimizer")
    optimizer = meta.MetaOptimizer(FLAGS.beta1, FLAGS.beta2, **net_config)
    if FLAGS.replicas > 1:
        from _eval_common import evaluate_replicas
        return evaluate_replicas(FLAGS, optimizer, problem, net_assignments, num_unrolls)
    meta_loss, _, _, step = optimizer.meta_loss(problem, FLAGS.unroll_len, net_assignments=net_assignments)
    _, update, reset, cost_op, _ = meta_loss

    with MonitoredSession() as sess:
        sess.run(reset)
        total_time = 0
        total_cost = 0
        loss_record = []
        for e in range(FLAGS.num_epochs):
            time, cost = util.run_eval_epoch(sess, cost_op, [update], num_unrolls, step=step,
                                             unroll_len=FLAGS.unroll_len)
            total_time += time
            total_cost += sum(cost) / num_unrolls
            loss_record += cost
        util.print_stats("Epoch {}".format(FLAGS.num_epochs), total_cost, total_time, FLAGS.num_epochs)

    if FLAGS.output_path is not None:
        if not os.path.exists(FLAGS.output_path):
            os.mkdir(FLAGS.output_path)
        output_file = "{}/{}_eval_loss_record.pickle-{}".format(FLAGS.output_path, FLAGS.optimizer, FLAGS.problem)
        with open(output_file, "wb") as l_record:
            pickle.dump([float(c) for c in loss_record], l_record)
        print("Saving evaluate loss record {}".format(output_file))


if __name__ == "__main__":
    main()
