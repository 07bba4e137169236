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


def test_config4_line_is_primary_when_sharded():
    """`bench.py --gpus N --config 4` (BASELINE configs[3]: Rastrigin d=100, global batch 1024 sharded over the GPUs)
    prints the config-4 line as the PRIMARY line: strong scaling, the global batch split evenly, every rank's copy of
    the all-reduced f(x_T) identical and equal to the one-process run on all 1024 problems (the
    first driver with an 8-GPU node gets the north-star number from one command)."""
    common = ["--config", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--unrolls-per-step", "1"]
    two = _bench(["--gpus", "2"] + common, {"L2O_BENCH_BACKEND": "gloo", "L2O_BENCH_ONE_DEVICE": "1"})
    cfg = two["config"]
    assert two["n_gpus"] == 2 and cfg["n_ranks_seen"] == 2 and cfg["backend"] == "gloo" and two["scaling"] == "strong"
    assert "Rastrigin d=100" in cfg["workload"] and "batch=512 per GPU (global 1024)" in cfg["workload"]
    assert cfg["baseline_config"] == "BASELINE.json configs[3]"
    ranks = two["final_loss_fx_T_per_rank"]
    assert len(ranks) == 2 and ranks[0] == ranks[1], ranks
    one = _bench(["--gpus", "1"] + common)
    assert one["config"]["baseline_config"].startswith("BASELINE.json configs[3]")
    assert two["final_loss_fx_T"] == pytest.approx(one["final_loss_fx_T"], rel=1e-6)
    assert two["final_loss_fx_T"] < two["fx_0"] / 5
