#!/usr/bin/env python
"""bench.py -- unroll-steps/sec of the L2O inner unroll loop on MI355X, through the
product API (open_l2o_amd.util.get_config -> MetaOptimizer.meta_loss -> one unroll per step).

Default workload (BASELINE.json configs[1]): L2O-DM (CoordinateWiseDeepLSTM, layers (20,20),
identity preprocess -- what util.get_config("quadratic") builds) on Quadratic d=128,
batch=128 per GPU, T=100 optimizer steps per unroll, fp32, synthetic data
(W, y ~ U[0,1), x0 ~ N(0, 0.01^2) as DM/problems.py:84-96; Sonnet-default random LSTM weights,
output Linear x0.1 so that the untrained optimizer's trajectory stays finite).

One "step" of this benchmark = one complete unroll: rewind x / LSTM state -> T x
{f(x), grad f, LSTM optimizer step, x += delta} -> f(x_T) -> per-step loss reduction
(-> all-reduce of the T+1 partial losses over ranks when N > 1).  Inputs are resident in
HBM when the timed region starts; nothing is copied to the host inside it.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --config 3 --steps 5          (the other BASELINE configs: --config 3 / 4 / 5)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  value = coordinate-steps per second, whole job:
    N_gpus * B_local * D * T * steps / wall_time.
Weak scaling: every GPU holds its own 128 problems of a global batch 128*N, the loss mean
is over the global batch (DM/problems.py:99), the only collective is the all-reduce of
T+1 floats per unroll.  The oracle is imported by the cpu_baseline leg only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK = 157.3e12


def alg_bytes_per_coord_step(problem, net, D, M):
    """x r+w + LSTM state r+w (+ RNNProp m, v r+w) + the optimizee's matrix
    streamed for the forward and for the gradient."""
    base = 8 + 640 + (16 if net == "rnnprop" else 0)
    if problem == "quadratic":
        return base + 8 * D + 4
    if problem == "lasso":
        return base + 8 * M + 4.0 * M / D
    if problem == "mnist":                       # minibatch [M x 784] read for forward and for gw1
        return base + 2.0 * M * 784 * 4 / D
    return base + 8 * D + 8                      # rastrigin


def alg_bytes_lasso_shared(net, B, D, M):
    """SURVEY.md 8(d), shared-A variant: the matrix is streamed twice per step for the WHOLE batch."""
    return 8 + 640 + (16 if net == "rnnprop" else 0) + 8.0 * M / B + 4.0 * M / D


def alg_flops_per_coord_step(problem, net, D, M):
    lstm = {"dm": 9800, "dm_logsign": 9960, "rnnprop": 12920 + 15}[net]
    if problem == "mnist":
        return lstm + 4 * M                      # 2 x 2 x batch MACs per weight
    return lstm + 4 * (M if problem == "lasso" else D)


def cpu_baseline(problem, net, arrays, weights, x0, T, max_seconds=20.0):
    """The reference path restated for the CPU (oracle/, test infrastructure), timed on this
    host's cores on the SAME inputs (whole unrolls, bounded to ~max_seconds): the plain-C +
    OpenMP port (oracle/l2o_oracle.c, one problem per thread) is the reported baseline; the
    NumPy oracle (multi-threaded BLAS gate matmuls) is timed once next to it for reference."""
    import oracle as O
    from oracle.c_oracle import c_unroll
    cfg = {"dm": O.DM_IDENTITY, "dm_logsign": O.DM_LOGSIGN, "rnnprop": O.RNNPROP}[net]
    B, M, D = arrays["W"].shape
    c_unroll(problem, cfg, weights, arrays, x0, 2)                # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    n = 0
    while True:
        fx, _, _, _, _, threads = c_unroll(problem, cfg, weights, arrays, x0, T)
        n += 1
        if time.perf_counter() - t0 > max_seconds or n >= 20:
            break
    dt = time.perf_counter() - t0
    out = {"value": B * D * T * n / dt, "unit": "coordinate-steps/s", "cores": int(threads),
           "host_cpus": os.cpu_count(), "kind": "port",
           "sample": "%d full unroll(s) of the same workload and inputs (C99+OpenMP port oracle/l2o_oracle.c, "
                     "%s/%s B=%d D=%d T=%d), %.1f s" % (n, net, problem, B, D, T, dt), "fx_T": float(fx[-1])}
    if problem == "quadratic" and B * D * T <= 2_000_000:
        prob = O.Quadratic(arrays["W"], arrays["y"])
        t0 = time.perf_counter()
        res = O.unroll(prob, cfg, weights, x0, O.net_initial_state(cfg, B * D), T)
        dt = time.perf_counter() - t0
        out["numpy_oracle"] = {"value": B * D * T / dt, "unit": "coordinate-steps/s",
                               "sample": "1 unroll, %.1f s" % dt, "fx_T": float(res.fx[-1])}
    return out


def build_workload(args, Bg):
    """Problem + optimizer through the product API."""
    from open_l2o_amd import meta, meta_rnnprop_eval, networks, util
    D, T = args.dims, args.unroll
    meta.set_random_seed(1234)                     # same global problem + weights on every rank
    opts = {"batch_size": Bg, "num_dims": D}
    if args.problem == "lasso":
        opts.update(l=0.1, num_rows=args.rows)
    if args.problem == "mnist":                    # not sharded: every GPU optimizes its own replica
        from open_l2o_amd import problems
        opts = {"batch_size": args.batch, "data": problems.synthetic_mnist(4096, seed=5)}
    problem, net_config, net_assignments = util.get_config(
        args.problem, problem_options=opts, net_name="RNNprop" if args.net == "rnnprop" else None)
    if args.problem == "lasso" and args.shared_matrix:
        # the shared-A variant of SURVEY.md 8(d): ONE sensing matrix [rows, dims] for every problem
        from open_l2o_amd import problems
        rng = np.random.default_rng(4321)
        rows = args.rows or D
        problem = problems.lasso_fixed(rng.random((rows, D), dtype=np.float32),
                                       rng.random((Bg, rows, 1), dtype=np.float32), l=0.1)
    if args.net == "dm_logsign" or (args.problem == "mnist" and args.net == "dm"):
        net_config = {"cw": util.get_default_net_config(None)}
    key = next(iter(net_config))
    cfg = dict(net_config[key])
    # Sonnet-default random init, output Linear x0.1 (see module docstring)
    weights = networks.factory(cfg["net"], cfg["net_options"]).variables
    weights = {m: {v: np.array(a) for v, a in d.items()} for m, d in weights.items()}
    weights["linear"] = {k: (a * np.float32(0.1)).astype(np.float32) for k, a in weights["linear"].items()}
    cfg["net_options"] = dict(cfg["net_options"], initializer=weights)
    net_config = {key: cfg}
    feed = {}
    if args.net == "rnnprop":
        optimizer = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **net_config)
        ml, _, _, step = optimizer.meta_loss(problem, T, net_assignments=net_assignments)
        feed = {step: 1}
    else:
        optimizer = meta.MetaOptimizer(**net_config)
        ml = optimizer.meta_loss(problem, T, net_assignments=net_assignments)
    return optimizer, ml, feed, weights


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dims", type=int, default=128)
    ap.add_argument("--batch", type=int, default=128, help="problems per GPU")
    ap.add_argument("--unroll", type=int, default=100, help="T")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--problem", default="quadratic", choices=["quadratic", "lasso", "rastrigin", "mnist"])
    ap.add_argument("--rows", type=int, default=None, help="lasso rows M (default: dims)")
    ap.add_argument("--shared-matrix", dest="shared_matrix", action="store_true",
                    help="lasso: ONE sensing matrix for all problems (SURVEY 8d shared-A variant)")
    ap.add_argument("--net", default="dm", choices=["dm", "dm_logsign", "rnnprop"])
    ap.add_argument("--config", type=int, default=None, choices=[2, 3, 4, 5],
                    help="preset = BASELINE.json configs[N-1] (per-GPU shard): 2 default; 3 RNNProp on Lasso 256x512, "
                         "batch 256, T=200; 4 DM on Rastrigin d=100, 128 problems per GPU, T=100; 5 RNNProp on the "
                         "MLP optimizee, minibatch 64, T=200 (forward unroll)")
    args = ap.parse_args()
    if args.config == 3:
        args.problem, args.net, args.dims, args.rows, args.batch, args.unroll = "lasso", "rnnprop", 512, 256, 256, 200
    elif args.config == 4:
        args.problem, args.net, args.dims, args.batch, args.unroll = "rastrigin", "dm", 100, 128, 100
    elif args.config == 5:
        args.problem, args.net, args.batch, args.unroll = "mnist", "rnnprop", 64, 200

    import torch
    import torch.distributed as dist
    from open_l2o_amd import _engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook: L2O_BENCH_BACKEND=gloo L2O_BENCH_ONE_DEVICE=1 runs the N > 1 code path with all
    # ranks on cuda:0 (a 1-GPU box cannot host two RCCL ranks); the driver never sets these
    backend = os.environ.get("L2O_BENCH_BACKEND", "nccl")
    if os.environ.get("L2O_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda:%d" % local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    eng = _engine.HipEngine("cuda:%d" % local_rank)        # raises without GPU / built extension
    _engine.set_default_engine(eng)

    D, B, T = args.dims, args.batch, args.unroll
    if args.problem == "mnist":                    # 784-20-10 MLP: 15 910 coordinates, `batch` = minibatch
        D = 784 * 20 + 20 + 20 * 10 + 10
    Bg = B * world
    optimizer, ml, feed, weights = build_workload(args, Bg)
    graph = optimizer.graph
    graph.reset()                                           # (first call: allocator / context warm-up)
    if eng.device.type == "cuda":
        torch.cuda.synchronize()
    t_reset = time.perf_counter()
    graph.reset()                                           # sample x0, W, y on the host (NumPy), upload this rank's shard
    if eng.device.type == "cuda":
        torch.cuda.synchronize()
    t_reset = time.perf_counter() - t_reset
    x0 = [v.value.clone() for v in graph.x]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    def one_unroll(i=None):
        graph.rewind(x0)                                    # x <- x0, LSTM state (m, v) <- 0
        # (the step-granular path replays its 2..6 x T small launches from a HIP graph)
        fx, _ = graph.launch(feed, commit=True, events=None if i is None else ev[i], use_graph=True)
        return fx

    def fence():
        graph.wait_fx()                                     # the asynchronous loss all-reduces (N > 1)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_unroll()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        fx = one_unroll(i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=eng.device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fx_host = eng.to_numpy(fx)
    eng.check_unroll_status()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    kern_ms_min = float(np.min([a.elapsed_time(b) for a, b in ev]))
    fused = graph.last_path == "fused"

    copy_gbps = None
    if rank == 0 and eng.device.type == "cuda":
        # achievable-copy figure of this box: device-to-device copy of 512 MiB, read + write counted
        src = torch.empty(128 << 20, dtype=torch.float32, device=eng.device).normal_()
        dst = torch.empty_like(src)
        for _ in range(2):
            dst.copy_(src)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(5):
            dst.copy_(src)
        c1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * src.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src, dst

    if rank == 0:
        coord_steps = (1 if args.problem == "mnist" else B) * D * T     # per GPU per unroll
        value = world * coord_steps * args.steps / dt
        Mrows = B if args.problem == "mnist" else (args.rows or D)
        shared = args.problem == "lasso" and args.shared_matrix
        bpc = (alg_bytes_lasso_shared(args.net, B, D, Mrows) if shared
               else alg_bytes_per_coord_step(args.problem, args.net, D, Mrows))
        alg = bpc * coord_steps                            # algorithmic bytes per unroll
        achieved = alg / (kern_ms * 1e-3)
        flops = alg_flops_per_coord_step(args.problem, args.net, D, Mrows) * coord_steps
        netname = {"dm": "L2O-DM CoordinateWiseDeepLSTM(20,20)", "dm_logsign": "L2O-DM (LogAndSign k=5)",
                   "rnnprop": "L2O-RNNProp (fc+ELU, tanh, 0.01)"}[args.net]
        probname = {"quadratic": "Quadratic d=%d" % D,
                    "lasso": "Lasso A in R^{%dx%d} l=0.1%s" % (Mrows, D, " (one A shared by the batch)" if shared else ""),
                    "rastrigin": "Rastrigin d=%d" % D,
                    "mnist": "MLP 784-20-10 (sigmoid) on synthetic MNIST-shaped data, minibatch %d" % B}[args.problem]
        is_c2 = (args.problem, args.net, D, B, T) == ("quadratic", "dm", 128, 128, 100)
        traffic, traffic_src = None, None
        for tag in ("c2", "c3"):                           # committed PMC passes of exactly this workload
            pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_%s.json" % tag)
            if not (os.path.exists(pmc_file) and fused and world == 1 and not shared):
                continue
            pmc = json.load(open(pmc_file))
            if pmc["workload"] == [args.problem, args.net, D, B, T] and (args.problem != "lasso" or Mrows == 256):
                traffic = pmc["traffic_bytes"]
                traffic_src = ("profiles/r01_pmc_%s.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                               "command, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950)" % tag)
        out = {
            "metric": "unroll-steps/sec (batch x params x T), %s on %s" % (netname.split(" ")[0], probname),
            "value": value, "unit": "coordinate-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s on %s, batch=%d per GPU (global %d), T=%d%s"
                                   % (netname, probname, B, Bg, T, ", BASELINE.json configs[1]" if is_c2 else ""),
                       "kernel": (("l2o_unroll, streaming form k_unroll_cu (one workgroup per problem, T steps in one "
                                   "launch, matrix streamed once per step, x / LSTM state / moments on-chip)"
                                   if D > 128 else
                                   "l2o_unroll (fused persistent, 2 CUs per problem when 2*batch <= #CUs)") if fused
                                  else "l2o_problem_fg + l2o_cwlstm_step per step"),
                       "arithmetic": "fp32 state, inputs and outputs; the LSTM gate GEMM is a 6-product 3-way bf16 "
                                     "split on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (fp32-level error, "
                                     "DESIGN.md 2); everything else fp32 VALU",
                       "api": "open_l2o_amd.util.get_config -> MetaOptimizer.meta_loss -> UnrollGraph.launch",
                       "parallelism": "problem-batch sharding x%d, all-reduce of T+1 floats" % world},
            "final_loss_fx_T": float(fx_host[-1]), "fx_0": float(fx_host[0]),
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                         "hbm_copy_measured_GBps": copy_gbps,
                         "reset_ms_host_sampling_plus_h2d": t_reset * 1e3,
                         "frac_of_measured_copy": None if not copy_gbps else achieved / 1e9 / copy_gbps,
                         "algorithmic_bytes_per_launch": alg, "alg_bytes_per_coord_step": bpc,
                         "kernel_ms_avg": kern_ms, "kernel_ms_min": kern_ms_min,
                         "fp32_tflops": flops / (kern_ms * 1e-3) / 1e12,
                         "fp32_frac_of_157.3TF": flops / (kern_ms * 1e-3) / FP32_PEAK,
                         "note": ("step-granular algorithmic bytes (SURVEY 8d) over the HIP-event time of the "
                                  "unroll kernels; the fused kernel keeps x, LSTM state and W on-chip, so real HBM "
                                  "traffic is far below this figure (frac can exceed 1) and the kernel is bound by one "
                                  "wave's serial instruction stream -- DESIGN.md 5") if not (fused and D > 128) else
                                 ("step-granular algorithmic bytes (SURVEY 8d: matrix twice + x / state / moments "
                                  "read and written per step) over the HIP-event time of the unroll kernel; the "
                                  "streaming form reads the matrix ONCE per step and keeps everything else on-chip, "
                                  "so frac can exceed 1 -- DESIGN.md 3.1c / 5")},
        }
        if world == 1 and not args.no_cpu_baseline and args.problem != "mnist" and not shared:
            names = {"quadratic": ("w", "y", None), "lasso": ("w", "y", None),
                     "rastrigin": ("A", "B", "C")}[args.problem]
            g = graph._by_name
            arrays = {"W": g[names[0]].eval(), "y": g[names[1]].eval().reshape(B, -1)}
            if names[2]:
                arrays["C"] = g[names[2]].eval().reshape(B, -1)
            arrays["l1"], arrays["alpha"] = 0.1, 10.0
            out["cpu_baseline"] = cpu_baseline(args.problem, args.net, arrays, weights,
                                               eng.to_numpy(x0[0]).reshape(B, D), T)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
