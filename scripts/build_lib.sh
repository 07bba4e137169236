#!/bin/bash
# A VARIANT of libl2o_hip.so with extra compiler flags (A/B and ablation builds): the same two translation units and flags as
# open_l2o_amd/csrc/Makefile (the kernels of csrc/l2o_ilp_kernels.h under max-ilp), the two compiles in parallel.
#   bash scripts/build_lib.sh build/var/lib_x.so -DL2O_SOMETHING=1 [more flags]      (l2o_build_id() of the result = the file's name)
set -e
OUT=$(realpath -m "$1"); shift
cd "$(dirname "$0")/../open_l2o_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize"
OBJ=$(dirname "$OUT")/obj_$(basename "$OUT" .so); mkdir -p "$OBJ"
/opt/rocm/bin/hipcc $FLAGS "$@" -DL2O_BUILD_ID="\"$(basename "$OUT" .so)\"" -c l2o_kernels.hip -o "$OBJ/main.o" &
/opt/rocm/bin/hipcc $FLAGS -Wno-unused-variable -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c l2o_kernels_ilp.hip -o "$OBJ/ilp.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,--no-undefined "$OBJ/main.o" "$OBJ/ilp.o" -o "$OUT"
rm -rf "$OBJ"
