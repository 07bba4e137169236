"""What scripts/evaluate_dm.py and scripts/evaluate_rnnprop.py share beyond the reference's flow: --replicas N."""
import os
import pickle
import time as _time

from open_l2o_amd import util
from open_l2o_amd.replicas import Replicas


def evaluate_replicas(FLAGS, optimizer, problem, net_assignments, num_unrolls):
    """N instances of the optimizee, each evaluated exactly like the single-instance flow (reset, then num_unrolls
    committed unrolls of unroll_len steps whose fx is recorded: DM/evaluate_dm.py:88-91, DM/evaluate_rnnprop.py:79-92,
    DM/util.py:78-89) -- as ONE unroll of num_unrolls * unroll_len steps per instance, all instances in launches of up to
    eight (Replicas.run: problems.mnist on the MI355X runs one instance per XCD)."""
    L = FLAGS.unroll_len
    reps = Replicas(optimizer, [problem] * FLAGS.replicas, num_unrolls * L, net_assignments=net_assignments)
    records = [[] for _ in range(FLAGS.replicas)]
    total_time = 0.0
    for e in range(FLAGS.num_epochs):
        reps.reset()
        t0 = _time.time()
        reps.run({reps.step: 1} if reps.step is not None else None)      # (RNNProp: the harness-fed step, DM/util.py:85-86)
        total_time += _time.time() - t0
        for j, fx in enumerate(reps.fx_arrays):
            records[j] += [float(fx[(k + 1) * L]) for k in range(num_unrolls)]       # the k-th unroll's fx
    mean_cost = sum(sum(r) for r in records) / (FLAGS.replicas * num_unrolls)
    util.print_stats("Epoch {} ({} replicas, kernel form: {})".format(FLAGS.num_epochs, FLAGS.replicas, reps.last_form),
                     mean_cost, total_time, FLAGS.num_epochs)
    print("final cost per replica: " + " ".join("%.5f" % r[-1] for r in records))
    if FLAGS.output_path is not None:
        if not os.path.exists(FLAGS.output_path):
            os.mkdir(FLAGS.output_path)
        output_file = "{}/{}_eval_loss_record.pickle-{}".format(FLAGS.output_path, FLAGS.optimizer, FLAGS.problem)
        with open(output_file, "wb") as l_record:
            pickle.dump(records, l_record)
        print("Saving evaluate loss record {}".format(output_file))
