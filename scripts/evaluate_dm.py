#!/usr/bin/env python
"""Learning 2 Learn evaluation -- the reference's DM/evaluate_dm.py re-hosted on
open_l2o_amd (same flags, same flow: get_config -> MetaOptimizer.meta_loss(problem, 1)
-> reset -> run_eval_epoch -> pickle the loss record).

    python scripts/evaluate_dm.py --problem=quadratic --num_steps=100 [--path=...] \
        [--output_path=out] [--unroll_len=1] [--batch_size=128 --num_dims=10]

``--unroll_len`` (ours, default 1 like DM/evaluate_dm.py:71) lets one session call run
several optimizer steps inside the fused kernel; the loss record then holds one value per call.
"""
import argparse
import logging
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from open_l2o_amd import meta, util  # noqa: E402
from open_l2o_amd.session import MonitoredSession  # noqa: E402


def main():
    flags = argparse.ArgumentParser()
    flags.add_argument("--optimizer", default="L2L", help="Optimizer.")
    flags.add_argument("--problem", default="simple", help="Type of problem.")
    flags.add_argument("--path", default=None, help="Path to saved meta-optimizer network.")
    flags.add_argument("--output_path", default=None, help="Path to output results.")
    flags.add_argument("--num_epochs", type=int, default=1, help="Number of evaluation epochs.")
    flags.add_argument("--num_steps", type=int, default=10000, help="Number of optimization steps per epoch.")
    flags.add_argument("--learning_rate", type=float, default=0.001, help="Learning rate.")
    flags.add_argument("--seed", type=int, default=None, help="Seed for the RNG.")
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
    problem, net_config, net_assignments = util.get_config(FLAGS.problem, FLAGS.path, problem_options=opts)

    if FLAGS.optimizer == "L2L":
        if FLAGS.path is None:
            logging.warning("Evaluating untrained L2L optimizer")
        optimizer = meta.MetaOptimizer(**net_config)
        if FLAGS.replicas > 1:
            from _eval_common import evaluate_replicas
            return evaluate_replicas(FLAGS, optimizer, problem, net_assignments, num_unrolls)
        meta_loss = optimizer.meta_loss(problem, FLAGS.unroll_len, net_assignments=net_assignments)
        _, update, reset, cost_op, _ = meta_loss
    else:
        raise ValueError("{} is not a valid optimizer".format(FLAGS.optimizer))

    with MonitoredSession() as sess:
        sess.run(reset)
        total_time = 0
        total_cost = 0
        loss_record = []
        for e in range(FLAGS.num_epochs):
            time, cost = util.run_eval_epoch(sess, cost_op, [update], num_unrolls)
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
