#!/bin/bash
# Round 3, lease 2: where the long-horizon drift comes from; the honest bench (fresh instance per unroll) on both two-CU
# forms; the trained-parity tests; what a process sees of a CU mask.
TAG=${1:-r03b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.."
set -x
timeout 600 python scripts/converged_step_accuracy.py --t 0,100,600 > $OUT/converged_step_accuracy.txt 2>&1
tail -30 $OUT/converged_step_accuracy.txt
for f in "" "L2O_PAIR_TWO_PASS=1"; do
  for c in 2 4 3; do
    [ "$c" = 3 ] && [ -n "$f" ] && continue
    env $f timeout 600 python bench.py --config $c --steps 20 --warmup 5 > $OUT/bench_c${c}_${f:-normal}.json 2> $OUT/bench_c${c}_${f:-normal}.err
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_c${c}_${f:-normal}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("config $c ${f:-normal}: value %.4g (replayed %.4g) ms/unroll %.4f kernel %.4f prepare %s fx0 %.4g fxT %.4g cpu %.3g relΔ %s" % (
        d["value"], d["value_replayed_problem"], d["ms_per_unroll"], r["kernel_ms_avg"], r.get("problem_prepare_ms"),
        d["fx_0"], d["final_loss_fx_T"], d.get("cpu_baseline", {}).get("value", 0), d.get("final_loss_rel_diff_vs_cpu_port")))
except Exception as e:
    print("config $c ${f:-normal}: FAILED", e)
PY
  done
done
timeout 1500 python -m pytest tests/test_trained_parity.py -x -q -m gpu -s > $OUT/pytest_trained.log 2>&1; tail -40 $OUT/pytest_trained.log
timeout 600 python -m pytest tests/test_generic_net.py tests/test_hip_kernels.py -x -q -m gpu -k "eager or preparation or two_cu_form or c4 or c2" > $OUT/pytest_misc.log 2>&1; tail -5 $OUT/pytest_misc.log
(./scripts/microbench/cu_mask_probe; echo "--- ROC_GLOBAL_CU_MASK=half"; ROC_GLOBAL_CU_MASK=0xffffffffffffffffffffffffffffffff ./scripts/microbench/cu_mask_probe; echo "--- HSA_CU_MASK=0:0-127"; HSA_CU_MASK=0:0-127 ./scripts/microbench/cu_mask_probe) > $OUT/cu_mask_probe.txt 2>&1
cat $OUT/cu_mask_probe.txt
