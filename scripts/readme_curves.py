#!/usr/bin/env python
"""The one external sanity check the reference offers for the LSTM optimizers: the convergence
curves its README publishes (/root/reference/README.md:34-42, Figs/ras.png = Figure 7 of the Open-L2O paper:
"generalized Rastrigin", n = 2 and n = 10, 10^3 iterations; every optimizer -- L2O-DM and L2O-RNNprop included --
plateaus at f ~ 5.5-6.2 for n = 2 (start ~ 22) and ~ 50-57 for n = 10 (start ~ 150); L2O-DM gets there within ~10
iterations, L2O-RNNprop within ~40 / ~150).  No numbers are published, only the figure: this is WEAK evidence, but it is
the only reference-held quantity that exercises a non-zero LSTM (the Sonnet cell itself is unpinned).

What this script does, with the repo's own drivers and the reference's schedule (DM/train_dm.py:39-44 defaults:
10 000 epochs of 100 steps, truncated BPTT 20, Adam 1e-3, evaluation every 100 epochs x 20 epochs, best kept;
util.get_config("rastrigin") = batch 128, DM/util.py:232-238):
  1. meta-train L2O-DM and L2O-RNNProp on problems.rastrigin(num_dims = 2 | 10),
  2. evaluate the saved optimizer for 1000 steps on fresh problem batches (DM/evaluate_dm.py: len_unroll 1,
     num_steps 1000), mean loss per iteration over --eval_epochs batches of 128 problems,
  3. write the curves (log-spaced iterations) and the README bands next to them as JSON.

    python scripts/readme_curves.py --out profiles/archive_r04/r04_readme_curves.json [--num_epochs 10000] [--max_seconds 90]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# read off Figs/ras.png (main panels + the zoomed insets); generous: the figure's own spread between optimizers
README_BANDS = {
    2: {"start": (18.0, 26.0), "plateau_at_1000": (5.0, 7.0), "source": "README.md:38 Figs/ras.png (a), inset 5.0-6.5"},
    10: {"start": (125.0, 165.0), "plateau_at_1000": (45.0, 62.0), "source": "README.md:38 Figs/ras.png (b), inset 40-65"},
}
ITERS = [1, 2, 3, 5, 10, 20, 30, 50, 100, 200, 300, 500, 1000]


def train(kind, n, save, num_epochs, max_seconds, seed):
    script = "train_rnnprop.py" if kind == "rnnprop" else "train_dm.py"
    cmd = [sys.executable, os.path.join(ROOT, "scripts", script), "--problem", "rastrigin", "--num_dims", str(n),
           "--num_epochs", str(num_epochs), "--evaluation_period", "100", "--evaluation_epochs", "20", "--num_steps", "100",
           "--unroll_length", "20", "--learning_rate", "0.001", "--seed", str(seed), "--save_path", save,
           "--max_seconds", str(max_seconds)]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    if out.returncode:
        raise RuntimeError(out.stdout[-2000:] + out.stderr[-2000:])
    evals = [l for l in out.stdout.splitlines() if l.startswith("epoch=")]
    return {"command": " ".join(cmd[1:]).replace(ROOT + "/", ""), "evaluations": len(evals),
            "last_eval": evals[-1] if evals else None,
            "total_time": [l for l in out.stdout.splitlines() if l.startswith("total time")][-1:]}


def evaluate(kind, n, path, steps, eval_epochs, seed):
    from open_l2o_amd import meta, meta_rnnprop_eval, util
    from open_l2o_amd.session import MonitoredSession
    meta.set_random_seed(seed)
    problem, net_config, assign = util.get_config("rastrigin", path, net_name="RNNprop" if kind == "rnnprop" else None,
                                                  problem_options={"num_dims": n})
    step = None
    if kind == "rnnprop":
        opt = meta_rnnprop_eval.MetaOptimizer(0.95, 0.95, **net_config)
        ml, _, _, step = opt.meta_loss(problem, 1, net_assignments=assign)
    else:
        opt = meta.MetaOptimizer(**net_config)
        ml = opt.meta_loss(problem, 1, net_assignments=assign)
    curves = []
    with MonitoredSession() as sess:
        for _ in range(eval_epochs):
            sess.run(ml.reset)
            _, cost = util.run_eval_epoch(sess, ml.fx, [ml.update], steps, step=step, unroll_len=1)
            curves.append(np.asarray(cost, np.float64))
    c = np.mean(curves, axis=0)                             # c[k - 1] = f after k optimizer steps
    return c, opt.graph.last_path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "archive_r04", "r04_readme_curves.json"))
    ap.add_argument("--num_epochs", type=int, default=10000)
    ap.add_argument("--max_seconds", type=float, default=90.0)
    ap.add_argument("--eval_epochs", type=int, default=10)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--kinds", default="dm,rnnprop")
    ap.add_argument("--dims", default="2,10")
    args = ap.parse_args()
    res = {"what": __doc__.split("\n\n")[0], "readme_bands": README_BANDS, "iterations": ITERS, "cases": {}}
    ok = True
    with tempfile.TemporaryDirectory() as tmp:
        for n in [int(v) for v in args.dims.split(",")]:
            for kind in args.kinds.split(","):
                save = os.path.join(tmp, "%s_%d" % (kind, n))
                tr = train(kind, n, save, args.num_epochs, args.max_seconds, seed=40 + n)
                key = "rp" if kind == "rnnprop" else "cw"
                curve, path = evaluate(kind, n, os.path.join(save, "%s.l2l-0" % key), args.steps, args.eval_epochs, seed=7)
                # f before any step: one more evaluation with the untrained path is not needed -- x0 ~ N(0, 1):
                # f(x0) is the first entry of a 0-step unroll; run_eval_epoch returns f AFTER each step, so take it apart
                band = README_BANDS[n]["plateau_at_1000"]
                inside = bool(band[0] <= curve[args.steps - 1] <= band[1]) if args.steps >= 1000 else None
                ok = ok and inside is not False
                res["cases"]["%s_rastrigin_n%d" % (kind, n)] = {
                    "training": tr, "kernel_path": path, "eval_problems": 128 * args.eval_epochs,
                    "f_after_k_steps": {str(k): float(curve[k - 1]) for k in ITERS if k <= len(curve)},
                    "f_at_1000_inside_readme_band": inside}
                print("%s rastrigin n=%d: f after 1 / 10 / 100 / 1000 steps = %s (README plateau %s): %s"
                      % (kind, n, ", ".join("%.3f" % curve[k - 1] for k in (1, 10, 100, 1000) if k <= len(curve)), band,
                         "inside" if inside else "OUTSIDE"), flush=True)
    res["all_inside_readme_bands"] = ok
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", args.out)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
