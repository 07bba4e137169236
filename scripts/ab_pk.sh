#!/bin/bash
# A/B of library variants over the three fused-unroll forms: bash scripts/ab_pk.sh OUTDIR
O=${1:-gpurun_out/abpk}; mkdir -p $O
run() { python bench.py --warmup 3 --no-cpu-baseline "$@" 2>>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-12s ONE_LDS=%-2s %-34s kernel_ms=%.4f  value=%.4g G fx_T=%r' % ('$LBL', '$L2O_ONE_LDS', '$*', r['kernel_ms_avg'], d['value']/1e9, d['final_loss_fx_T']))" | tee -a $O/ab.txt; }
for rep in 1 2; do
for v in build/var/lib_*.so; do
  export L2O_HIP_LIB=$PWD/$v; LBL=$(basename $v .so)
  unset L2O_ONE_LDS
  run --steps 20
  run --config 4 --steps 6
  run --batch 256 --steps 10
  L2O_ONE_LDS=3 run --config 4 --steps 6
  L2O_ONE_LDS=3 run --batch 256 --steps 10
done
done
