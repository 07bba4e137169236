#!/bin/bash
# Register / spill counts of a FEW kernel instantiations in seconds instead of a 3-minute library build: cuts
# l2o_kernels.hip behind the header that defines the kernel, appends explicit instantiations, compiles device-only to
# assembly and prints vgpr / spill / scratch from the metadata.  (How the spills of k_unroll_cu8 were tracked down.)
#   bash scripts/tu_regs.sh l2o_unroll_cu8.h 'l2o::k_unroll_cu8<2, 2, 4, false>(UnrollArgs)' ['...more...'] [-- extra hipcc flags]
# (the kernels of csrc/l2o_ilp_kernels.h ship under max-ilp: add  -- -mllvm -amdgpu-sched-strategy=max-ilp  to see what ships)
set -e
cd "$(dirname "$0")/.."
HDR=$1; shift
INST=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do INST+=("$1"); shift; done; [ "$1" = "--" ] && shift
mkdir -p build
n=$(grep -n "#include \"$HDR\"" open_l2o_amd/csrc/l2o_kernels.hip | cut -d: -f1)
head -n $n open_l2o_amd/csrc/l2o_kernels.hip > build/tu_regs.hip
sed -i 's|#include "l2o_|#include "../open_l2o_amd/csrc/l2o_|; s|#include "../../include|#include "../include|' build/tu_regs.hip
for i in "${INST[@]}"; do echo "template __global__ void $i;" >> build/tu_regs.hip; done
grep -q "^#ifndef L2O_TU_ILP" build/tu_regs.hip && echo "#endif" >> build/tu_regs.hip     # (the cut lies inside the main-TU-only part)
cd build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize \
  --cuda-device-only "$@" -S tu_regs.hip -o tu_regs.s 2>&1 | grep -E "error" -A3 | head -12
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size):" tu_regs.s | paste - - - - | \
  awk '{print $2, "scratch", $4, "vgpr", $6, "spill", $8}'
