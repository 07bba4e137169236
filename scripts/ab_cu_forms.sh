#!/bin/bash
# A/B of the streaming fused unroll's forms (L2O_UNROLL_CU = 1 eight waves, 4 register tiles | 3 eight waves, 3 register tiles | 2 four waves)
O=${1:-gpurun_out/cu}; mkdir -p $O
run() { python bench.py --warmup 2 --no-cpu-baseline "$@" 2>>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('UNROLL_CU=%-2s %-40s kernel_ms=%.4f  value=%.4g G fx_T=%r' % ('$L2O_UNROLL_CU', '$*', r['kernel_ms_avg'], d['value']/1e9, d['final_loss_fx_T']))" | tee -a $O/cu_forms.txt; }
for rep in 1 2; do
for f in 2 1 3; do
  export L2O_UNROLL_CU=$f
  run --config 3 --steps 4
done
done
for f in 2 1; do
  export L2O_UNROLL_CU=$f
  run --problem lasso --net dm --untrained --dims 512 --rows 256 --batch 256 --unroll 100 --steps 4
  run --problem lasso --net rnnprop --untrained --dims 256 --rows 128 --batch 256 --unroll 100 --steps 4
done
