"""The N > 1 path of bench.py under the driver's GPU test even without a multi-GPU node: two ranks
(gloo, both on cuda:0 -- a 1-GPU box cannot host two RCCL ranks) shard the global batch; every rank must end with the
SAME all-reduced loss, and that loss must equal the single-process run on the whole global batch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_two_ranks_one_device_equal_the_single_process_global_batch():
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-also", "--unrolls-per-step", "2"]
    two = _bench(["--gpus", "2"] + common, {"L2O_BENCH_BACKEND": "gloo", "L2O_BENCH_ONE_DEVICE": "1"})
    assert two["n_gpus"] == 2 and two["config"]["n_ranks_seen"] == 2 and two["config"]["backend"] == "gloo"
    ranks = two["final_loss_fx_T_per_rank"]
    assert len(ranks) == 2 and ranks[0] == ranks[1], ranks                  # one all-reduce, identical on every rank
    one = _bench(["--gpus", "1", "--batch", "256"] + common)                 # the same global batch of 256 in one process
    a, b = two["final_loss_fx_T"], one["final_loss_fx_T"]
    print("2 ranks x 128 problems: fx_T %.9g; 1 rank x 256: %.9g" % (a, b))
    assert two["fx_0"] == pytest.approx(one["fx_0"], rel=1e-6)
    assert a == pytest.approx(b, rel=1e-6)
    assert a < two["fx_0"] / 5                                                 # (the trained optimizer: the loss falls)
