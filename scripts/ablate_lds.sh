#!/bin/bash
# Timing ablations of k_unroll_lds (build/var/lib_*.so from scripts/build_variants.sh with the L2O_LDS_ABL_* macros):
#   bash scripts/ablate_lds.sh OUTDIR       -> OUTDIR/ablate_lds.txt
# Quadratic d = 128, 256 problems (one per CU), T = 100: what the step time becomes without the barriers / the fragment
# reads / the GEMV operand reads (wrong numerics, timing only), and with ONE wave per SIMD (d = 64: 4 tiles).
O=${1:-gpurun_out/abl}; mkdir -p $O
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
T=100; cyc=r['kernel_ms_avg']*1e-3*2.4e9/(T+1)
print('%-28s %-22s kernel_ms=%.4f  cycles/step(2.4GHz)=%.0f  value=%.4g G fx_T=%r' % ('$LBL', '$*', r['kernel_ms_avg'], cyc, d['value']/1e9, d['final_loss_fx_T']))" | tee -a $O/ablate_lds.txt; }
export L2O_ONE_LDS=2
for v in build/var/lib_*.so; do
  export L2O_HIP_LIB=$PWD/$v; LBL=$(basename $v .so)
  run --batch 256
  case $LBL in *anynw*) run --batch 256 --dims 64; run --batch 256 --dims 80; run --batch 256 --dims 96; run --batch 256 --dims 112;; esac
done
