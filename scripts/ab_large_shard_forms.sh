#!/bin/bash
# A/B of the large-shard forms of the fused unroll (L2O_ONE_LDS = 0 chunked two-CU | 1/3 k_unroll_pair2 | 2 k_unroll_lds):
#   bash scripts/ab_large_shard_forms.sh OUTDIR
O=${1:-gpurun_out/forms}; mkdir -p $O
run() { python bench.py --warmup 3 --no-cpu-baseline "$@" 2>>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ONE_LDS=%s %-38s kernel_ms=%.4f  value=%.4g G fx_T=%r  [%s]' % ('$L2O_ONE_LDS', '$*', r['kernel_ms_avg'], d['value']/1e9, d['final_loss_fx_T'], r['kernel'][:40]))" | tee -a $O/large_shard_forms.txt; }
for rep in 1 2; do
for f in 0 2 3; do
  export L2O_ONE_LDS=$f
  run --config 4 --steps 6
  run --batch 256 --steps 10
  run --batch 512 --steps 6 --unrolls-per-step 8
  run --batch 1024 --steps 4 --unrolls-per-step 4
done
L2O_ONE_LDS=3 run --batch 128 --steps 10
L2O_ONE_LDS=0 run --batch 128 --steps 10
done
