#!/usr/bin/env python
"""bench.py -- unroll-steps/sec of the L2O inner unroll loop on MI355X.

Workload (BASELINE.json configs[1]): L2O-DM (CoordinateWiseDeepLSTM, layers (20,20),
identity preprocess -- what util.get_config("quadratic") builds) on Quadratic d=128,
batch=128 per GPU, T=100 optimizer steps per unroll, fp32, synthetic data
(W, y ~ U[0,1), x0 ~ N(0, 0.01^2), Sonnet-default random LSTM weights).

One "step" of this benchmark = one complete unroll: reset x/LSTM state -> T x
{f(x), grad f, LSTM optimizer step, x += delta} -> f(x_T) -> per-step loss reduction
(-> all-reduce of the T+1 partial losses over ranks when N > 1).  Inputs are resident
in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  value = coordinate-steps per second, whole job:
    N_gpus * B_local * D * T * steps / wall_time.
Weak scaling: every GPU holds its own 128 problems, the loss mean is over the global
batch 128*N (DM/problems.py:99), the only collective is the all-reduce of T+1 floats.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# algorithmic bytes per coordinate-step, SURVEY.md section 8(d): x r+w (8) + LSTM state
# r+w (640) + optimizee row+column+y (8*D + 4)
def alg_bytes_per_coord_step(D):
    return 8 + 640 + 8 * D + 4


HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK = 157.3e12


def cpu_baseline(D, B, T, max_seconds=25.0):
    """The NumPy fp32 oracle (the CPU restatement of the reference path) timed on this
    host: whole unrolls of the same workload until ~max_seconds are spent."""
    import oracle as O
    from helpers import make_params, make_problem
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count()
    cfg = O.DM_IDENTITY
    params = make_params(cfg, seed=0, trained_like=True)
    prob, x0, _ = make_problem("quadratic", B, D, seed=1)
    st0 = O.net_initial_state(cfg, B * D)
    O.unroll(prob, cfg, params, x0, st0, 2)            # warm-up
    t0 = time.perf_counter()
    n = 0
    fxT = None
    while True:
        res = O.unroll(prob, cfg, params, x0, st0, T)
        fxT = float(res.fx[-1])
        n += 1
        if time.perf_counter() - t0 > max_seconds or n >= 3:
            break
    dt = time.perf_counter() - t0
    return {"value": B * D * T * n / dt, "unit": "coordinate-steps/s", "cores": int(threads),
            "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "%d full unroll(s) of the same workload (NumPy fp32 oracle, B=%d D=%d T=%d), %.1f s"
                      % (n, B, D, T, dt), "fx_T": fxT}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dims", type=int, default=128)
    ap.add_argument("--batch", type=int, default=128, help="problems per GPU")
    ap.add_argument("--unroll", type=int, default=100, help="T")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import oracle as O
    from helpers import device_problem, make_params, make_problem, spec_of
    from open_l2o_amd._engine import HipEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda:%d" % local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    eng = HipEngine("cuda:%d" % local_rank)

    D, B, T = args.dims, args.batch, args.unroll
    Bg = B * world
    cfg = O.DM_IDENTITY
    spec = spec_of(cfg)
    # random-init Sonnet-default weights (output Linear x0.1 so that the untrained optimizer
    # takes small steps and the trajectory stays finite); same weights on every rank
    params = make_params(cfg, seed=0, trained_like=True)
    prob, x0, arrays = make_problem("quadratic", B, D, seed=1 + rank)   # rank's own problems
    wpack = eng.pack_weights(spec, params)
    pd = device_problem(eng, arrays, B, D, B_global=Bg)
    fused = eng.unroll_supported(spec, pd)
    x0d = eng.tensor(x0)
    x, st = eng.empty(B, D), eng.state_alloc(B, D)
    fx_part, fx = eng.zeros((T + 1) * B), eng.zeros(T + 1)
    f1, g = eng.zeros(B), eng.zeros(B, D)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    def one_unroll(i=None):
        x.copy_(x0d)                                       # reset (DM/meta.py:379-383)
        st.zero_()
        if i is not None:
            ev[i][0].record()
        if fused:
            eng.unroll(spec, wpack, pd, x, st, None, None, T, 1, fx_part)
        else:
            for t in range(T):
                eng.problem_fg(pd, x, fx_part[t * B:(t + 1) * B], g)
                eng.lstm_step(spec, wpack, g, None, None, 0.0, 0.0, st, x, B, D)
            eng.problem_fg(pd, x, fx_part[T * B:(T + 1) * B], None)
        if i is not None:
            ev[i][1].record()
        eng.reduce_fx(fx_part, T + 1, B, Bg, fx)
        if world > 1:
            dist.all_reduce(fx)                            # sum of per-rank partial means

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_unroll()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_unroll(i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=eng.device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fx_host = eng.to_numpy(fx)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    kern_ms_min = float(np.min([a.elapsed_time(b) for a, b in ev]))

    if rank == 0:
        coord_steps = B * D * T                            # per GPU per unroll
        value = world * coord_steps * args.steps / dt
        alg = alg_bytes_per_coord_step(D) * coord_steps    # algorithmic bytes per launch
        achieved = alg / (kern_ms * 1e-3)
        flops = (9800 + 4 * D) * coord_steps
        out = {
            "metric": "unroll-steps/sec (batch x params x T), L2O-DM on Quadratic d=%d" % D,
            "value": value, "unit": "coordinate-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "L2O-DM CoordinateWiseDeepLSTM(20,20) on Quadratic d=%d, batch=%d per GPU "
                                   "(global %d), T=%d, BASELINE.json configs[1]" % (D, B, Bg, T),
                       "kernel": "k_unroll (fused persistent)" if fused else "k_problem_fg + k_cwlstm_step per step",
                       "parallelism": "problem-batch sharding x%d, all-reduce of T+1 floats" % world},
            "final_loss_fx_T": float(fx_host[-1]), "fx_0": float(fx_host[0]),
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": None,
                         "alg_bytes_per_coord_step": alg_bytes_per_coord_step(D),
                         "kernel_ms_avg": kern_ms, "kernel_ms_min": kern_ms_min,
                         "fp32_tflops": flops / (kern_ms * 1e-3) / 1e12,
                         "fp32_frac_of_157.3TF": flops / (kern_ms * 1e-3) / FP32_PEAK,
                         "note": "step-granular algorithmic bytes (SURVEY 8d); the fused kernel keeps x, LSTM "
                                 "state and W on-chip, so real HBM traffic is far below this figure and the "
                                 "kernel is matrix-core/VALU bound -- see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(D, B, T)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
