#!/bin/bash
# builds library variants into build/var in parallel: bash scripts/build_variants.sh name:-DFLAG[,-DFLAG2] ...
# (lib_a_base.so = the in-tree build); A/B them on the GPU with scripts/variants.sh
cd "$(dirname "$0")/../open_l2o_amd/csrc"
mkdir -p ../../build/var; rm -f ../../build/var/lib_*.so
make 2>&1 | grep -E " error"; cp ../libl2o_hip.so ../../build/var/lib_a_base.so
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}; f=${f//,/ }
  bash ../../scripts/build_lib.sh ../../build/var/lib_$n.so $f 2>&1 | grep -E " error" &
done
wait
ls ../../build/var
