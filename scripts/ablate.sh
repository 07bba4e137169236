#!/bin/bash
# Timing ablations of the fused unroll kernel (results are WRONG by construction; only
# kernel time is read).  Build variants here (CPU box), run on the GPU via gpurun:
#   bash scripts/ablate.sh build ; gpurun -- bash scripts/ablate.sh run
set -e
cd "$(dirname "$0")/.."
VARS=${VARS:-"BASE MFMA TRANS GEMV BARRIER MFMA_TRANS EXCHANGE EXCHANGE_BARRIER"}
if [ "$1" = build ]; then
  mkdir -p build/ablate
  for v in $VARS; do
    defs=""
    for d in ${v//_/ }; do [ "$d" != BASE ] && defs="$defs -DL2O_ABLATE_$d"; done
    bash scripts/build_lib.sh build/ablate/lib_$v.so $defs &
  done
  wait; ls -la build/ablate
else
  for v in $VARS; do
    echo -n "$v: "
    L2O_HIP_LIB=$PWD/build/ablate/lib_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also "${@:2}" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms_avg=%.4f min=%.4f' % (d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min']))"
  done
fi
