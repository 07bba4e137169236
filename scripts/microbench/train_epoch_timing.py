"""Wall time of meta-training EPOCHS the way the reference's drivers run them (DM/train_dm.py: num_steps = 100,
unroll_length = 20 -> util.run_epoch = reset + 5 training unrolls): per epoch and per unroll, with this round's host-side
changes switched off one by one.
python scripts/microbench/train_epoch_timing.py [B D T unrolls epochs]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(B, D, T, U, E):
    import torch
    from open_l2o_amd import meta, util
    from open_l2o_amd.session import Session
    meta.set_random_seed(3)
    problem, net_config, assignments = util.get_config("quadratic", problem_options={"batch_size": B, "num_dims": D})
    opt = meta.MetaOptimizer(**net_config)
    step, update, reset, fx, x = opt.meta_minimize(problem, T, learning_rate=1e-3, net_assignments=assignments)
    with Session() as sess:
        for _ in range(3):
            util.run_epoch(sess, fx, [update, step], reset, U)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(E):
            _, cost = util.run_epoch(sess, fx, [update, step], reset, U)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / E
    print("%-46s epoch %.3f ms = %.3f ms per unroll (last cost %.4g)" % (os.environ.get("L2O_LABEL", ""), dt * 1e3, dt * 1e3 / U, cost))


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [128, 128, 20, 5, 40]
    if os.environ.get("L2O_LABEL") is not None:
        one(*a)
    else:
        print("B=%d D=%d T=%d, %d unrolls per epoch, %d epochs" % tuple(a))
        for label, env in (("default", {}), 
                           ("one sync per unroll (L2O_NO_DEFER=1)", {"L2O_NO_DEFER": "1"}),
                           ("one sync per unroll, again", {"L2O_NO_DEFER": "1"})):
            e = dict(os.environ, L2O_LABEL=label, **env)
            subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(v) for v in a], env=e, check=False)
