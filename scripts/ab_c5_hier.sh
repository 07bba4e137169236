#!/bin/bash
# A/B of l2o_mlp_unroll's all-reduce protocols on config 5: the in-tree build with the XCD-hierarchical protocol on / off
# (L2O_NO_MLP_HIER=1), and every build/var/lib_*.so (e.g. one compiled with -DL2O_MU_NO_HIER)
run() { python bench.py --config 5 --steps 5 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s value=%.4g G  kernel_ms avg=%.4f  fx_T=%r' % ('$1', d['value']/1e9, r['kernel_ms_avg'], d['final_loss_fx_T']))"; }
for rep in 1 2; do
  unset L2O_HIP_LIB L2O_NO_MLP_HIER; run "in-tree hier=on"
  L2O_NO_MLP_HIER=1 run "in-tree hier=off(runtime)"
  for v in build/var/lib_*.so; do L2O_HIP_LIB=$PWD/$v run "$(basename $v .so)"; done
done
