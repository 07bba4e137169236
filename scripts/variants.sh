#!/bin/bash
# A/B timing of library variants (build/var/lib_*.so, built on the CPU box): bash scripts/variants.sh OUTDIR [bench args]
O=${1:-gpurun_out/var}; shift
mkdir -p $O
for v in build/var/lib_*.so; do
  n=$(basename $v .so)
  for rep in 1 2; do
    L2O_HIP_LIB=$PWD/$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also "$@" 2>$O/$n.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-18s value=%.4g G  kernel_ms avg=%.4f min=%.4f  fx_T=%r' % ('$n', d['value']/1e9, r['kernel_ms_avg'], r['kernel_ms_min'], d['final_loss_fx_T']))" | tee -a $O/variants.txt
  done
done
