#!/bin/bash
# where k_atb_bx3's time goes: ablations (-DL2O_ATB_ABLATE, binaries built by hand into build/atb_ablate/) and workgroups per CU
TAG=${1:-r03q3}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
R=1638400
(for v in 0 1 2 3 4 5; do [ -x build/atb_ablate/ab$v ] && { echo "== L2O_ATB_ABLATE=$v"; timeout 120 build/atb_ablate/ab$v $R 1 | grep -v "k_atb  "; }; done
 [ -x build/atb_ablate/ab5 ] && { echo "== L2O_ATB_ABLATE=5, 1 workgroup per CU"; timeout 120 build/atb_ablate/ab5 $R 1 1 | grep -v "k_atb  "; }
 echo "== mask 2 (RNNProp, 103 x 181)"; timeout 120 build/atb_ablate/ab0 $R 2 | grep bx3) 2>&1 | tee $O/atb_bx3_ablation.txt
