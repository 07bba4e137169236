# config-3 A/B of variant libraries: bash scripts/debug/c3_ab.sh LIB.so [LIB.so ...]   ("-" = the shipped library)
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset L2O_HIP_LIB; else export L2O_HIP_LIB=$PWD/$lib; fi
  for rep in 1 2; do
    timeout 300 python bench.py --config 3 --steps 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('$lib', '%.3f G' % (d['value'] / 1e9), 'ms/unroll %.3f' % d['ms_per_unroll'], 'hbm frac %.3f' % d['roofline']['frac'], 'diff vs cpu', d.get('final_loss_rel_diff_vs_cpu_port'))"
  done
done
