for v in A B C D; do
  lib=build/libl2o_hip_frag$v.so
  for cfgargs in "5 --replicas 8" "3" "4" "2 --batch 256"; do
    set -- $cfgargs; c=$1; shift
    if [ "$c" = 2 ]; then extra="$@ --steps 5"; else extra="--config $c $@ --steps 5"; fi
    L2O_HIP_LIB=$PWD/$lib timeout 300 python bench.py $extra --min-timed-seconds 0.3 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('frag$v [$cfgargs]: value %.3f G, ms/unroll %.4f, kernel_ms %.4f, %s, fx_T %.6g' % (d['value']/1e9, d['ms_per_unroll'], r['kernel_ms_avg'], r['kernel'][:14], d['final_loss_fx_T']))"
  done
done
