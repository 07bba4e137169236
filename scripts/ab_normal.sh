run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$TAGN value=%.4g kernel_ms avg=%.4f min=%.4f fx_T=%r' % (d['value']/1e9, r['kernel_ms_avg'], r['kernel_ms_min'], d['final_loss_fx_T']))"; }
for i in 1 2; do TAGN=normal run "$@"; done
for i in 1 2; do TAGN=twopass L2O_PAIR_TWO_PASS=1 run "$@"; done
