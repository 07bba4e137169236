#!/bin/bash
# A/B of k_unroll_lds build variants (build/var/lib_*.so): bash scripts/ab_lds_variants.sh OUTDIR
O=${1:-gpurun_out/ablds}; mkdir -p $O
run() { python bench.py --warmup 3 --no-cpu-baseline "$@" 2>>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-20s %-34s kernel_ms=%.4f  value=%.4g G fx_T=%r' % ('$LBL', '$*', r['kernel_ms_avg'], d['value']/1e9, d['final_loss_fx_T']))" | tee -a $O/ab_lds_variants.txt; }
for rep in 1 2; do
for v in build/var/lib_*.so; do
  export L2O_HIP_LIB=$PWD/$v; LBL=$(basename $v .so)
  run --batch 256 --steps 10
  run --config 4 --steps 6
done
done
