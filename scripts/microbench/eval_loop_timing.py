"""Steady-state wall time of the PRODUCT evaluation path -- Session.run([fx, update]): an in-place committed unroll per
call, status check and (round 5) the recovery snapshot included -- at config-2 shape; L2O_NO_RECOVERY=1 shows what the
snapshot costs.   python scripts/microbench/eval_loop_timing.py [B D T [N]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open_l2o_amd import meta, util
from open_l2o_amd.session import Session

B, D, T = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (128, 128, 100)))
N = int(sys.argv[4]) if len(sys.argv) > 4 else 300
meta.set_random_seed(3)
problem, net_config, assignments = util.get_config("quadratic", problem_options={"batch_size": B, "num_dims": D})
opt = meta.MetaOptimizer(**net_config)
ml = opt.meta_loss(problem, T, net_assignments=assignments)
with Session() as sess:
    sess.run(ml.reset)
    for _ in range(20):
        sess.run([ml.fx, ml.update])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        sess.run([ml.fx, ml.update])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
print("Session.run([fx, update]) B=%d D=%d T=%d: %.4f ms per unroll (%.3f G coordinate-steps/s), recovery %s, kernel %s"
      % (B, D, T, dt * 1e3, B * D * T / dt / 1e9, "off" if os.environ.get("L2O_NO_RECOVERY") else "on",
         opt.graph.engine.last_unroll_form()[0]))
