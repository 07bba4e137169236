#!/bin/bash
# RNNProp on large shards of a d <= 128 optimizee: chunked two-CU form (L2O_ONE_LDS=0) vs k_unroll_lds (default)
O=${1:-gpurun_out/rp}; mkdir -p $O
run() { python bench.py --warmup 3 --no-cpu-baseline --net rnnprop --untrained "$@" 2>>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ONE_LDS=%-2s %-46s kernel_ms=%.4f  value=%.4g G fx_T=%r  [%s]' % ('$L2O_ONE_LDS', '$*', r['kernel_ms_avg'], d['value']/1e9, d['final_loss_fx_T'], r['kernel'][:36]))" | tee -a $O/rnnprop_large_shards.txt; }
for rep in 1 2; do
for f in 0 1; do
  export L2O_ONE_LDS=$f
  run --batch 256 --steps 10
  run --batch 1024 --steps 4 --unrolls-per-step 4
  run --problem rastrigin --dims 100 --batch 1024 --steps 4 --unrolls-per-step 4
done
done
