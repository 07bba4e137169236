#!/bin/bash
# k_unroll_cu8: ring depth (library variants build/var/lib_*.so) x register tiles (L2O_UNROLL_CU = 3: four, 4: three + one LDS slot)
O=${1:-gpurun_out/cu8ring}; mkdir -p $O
run() { python bench.py --warmup 2 --no-cpu-baseline --config 3 --steps 4 2>>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-12s UNROLL_CU=%-2s kernel_ms=%.4f  value=%.4g G fx_T=%r' % ('$LBL', '$L2O_UNROLL_CU', r['kernel_ms_avg'], d['value']/1e9, d['final_loss_fx_T']))" | tee -a $O/cu8_ring.txt; }
for rep in 1 2; do
for v in build/var/lib_*.so; do
  export L2O_HIP_LIB=$PWD/$v; LBL=$(basename $v .so)
  for f in 3 4; do export L2O_UNROLL_CU=$f; run; done
done
L2O_UNROLL_CU=2 LBL=four_wave run
done
