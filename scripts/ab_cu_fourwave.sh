#!/bin/bash
# four-wave streaming kernel (L2O_UNROLL_CU=2) after the ring-liveness fix: ring 3 (in-tree) vs ring 4 (variant), config 3
O=${1:-gpurun_out/cu4}; mkdir -p $O
export L2O_UNROLL_CU=2
for v in build/var/lib_*.so; do
  for rep in 1 2; do
  L2O_HIP_LIB=$PWD/$v python bench.py --warmup 2 --no-cpu-baseline --config 3 --steps 4 2>>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-14s four-wave kernel_ms=%.4f  value=%.4g G fx_T=%r' % ('$(basename $v .so)', r['kernel_ms_avg'], d['value']/1e9, d['final_loss_fx_T']))" | tee -a $O/cu_fourwave.txt
  done
done
