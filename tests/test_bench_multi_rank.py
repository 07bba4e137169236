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


def test_eight_gloo_ranks_one_device_rehearse_the_config4_line():
    """The command the first driver with an 8-GPU node will run, `bench.py --gpus 8 --config 4`, end to end with EIGHT ranks
    (gloo, all on cuda:0): the primary line is BASELINE configs[3] (1024 Rastrigin problems, 128 per rank, strong
    scaling), all eight ranks hold the same all-reduced loss, the line carries `roofline` and a rank-0-only, bounded
    `cpu_baseline`, and the whole command finishes in two minutes."""
    import time
    t0 = time.time()
    line = _bench(["--gpus", "8", "--config", "4", "--steps", "2", "--warmup", "1", "--unrolls-per-step", "2", "--no-also"],
                  {"L2O_BENCH_BACKEND": "gloo", "L2O_BENCH_ONE_DEVICE": "1"})
    secs = time.time() - t0
    cfg = line["config"]
    assert line["n_gpus"] == 8 and cfg["n_ranks_seen"] == 8 and line["scaling"] == "strong"
    assert cfg["baseline_config"] == "BASELINE.json configs[3]"
    assert "Rastrigin d=100" in cfg["workload"] and "batch=128 per GPU (global 1024)" in cfg["workload"]
    ranks = line["final_loss_fx_T_per_rank"]
    assert len(ranks) == 8 and len(set(ranks)) == 1, ranks
    assert line["final_loss_fx_T"] < line["fx_0"] / 5                         # (the trained optimizer: the loss falls)
    roof, cpu = line["roofline"], line["cpu_baseline"]
    assert roof["bound"] == "valu_pipe" and 0.0 < roof["frac"] < 1.0 and roof["kernel"].startswith("k_unroll_pair")
    assert cpu["kind"] == "port" and cpu["value"] > 0 and 1 <= cpu["cores"] <= (os.cpu_count() or 1)
    assert "rank 0" in cpu["sample"]
    one = _bench(["--gpus", "1", "--config", "4", "--steps", "2", "--warmup", "1", "--unrolls-per-step", "2",
                  "--no-cpu-baseline", "--no-also"])                           # (same launch count: the same instance of the ring)
    assert line["final_loss_fx_T"] == pytest.approx(one["final_loss_fx_T"], rel=1e-6)
    print("8 gloo ranks on one device: %.1f s, fx_T %.9g, cpu leg: %s" % (secs, line["final_loss_fx_T"], cpu["sample"]))
    assert secs < 120.0, secs


def test_stdout_is_one_json_line_even_with_a_real_rccl_communicator():
    """The driver reads ONE JSON line from stdout.  RCCL prints a version banner through C stdio when its first
    communicator is created (seen in round 6's first lease: five lines BEHIND the JSON line): bench.py keeps fd 1 for its
    line alone.  The run itself is config 4's shard of 8 with its all-reduce issued through a world-size-1 RCCL group."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--emulate-world", "8", "--real-collective",
                        "--steps", "2", "--warmup", "1", "--unrolls-per-step", "2", "--no-cpu-baseline", "--no-also"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["config"]["collective"]["backend"] == "nccl" and line["config"]["collective"]["world_size"] == 1
    assert line["roofline"]["kernel"].startswith("k_unroll_pair")
