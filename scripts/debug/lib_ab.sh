# Alternating A/B of a variant library against the shipped one over the default bench line (all configs):
#   bash scripts/debug/lib_ab.sh build/libl2o_hip_VARIANT.so [REPEATS]
V=$1; N=${2:-2}
for rep in $(seq $N); do
  for lib in "$V" ""; do
    if [ -n "$lib" ]; then export L2O_HIP_LIB=$PWD/$lib; else unset L2O_HIP_LIB; fi
    timeout 500 python bench.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('%-34s' % ('$lib' or 'shipped'), ' '.join('%s %.3f' % (k.replace('_value', ''), d[k] / 1e9) for k in ('value', 'c3_value', 'c4_value', 'c4s8_value', 'c4s8rccl_value', 'c5_value', 'c5x8_value')))"
  done
done
