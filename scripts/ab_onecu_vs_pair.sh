#!/bin/bash
# A/B of the large-shard forms of the fused unroll: consecutive chunk launches of the two-CU kernel (default until round 4),
# the one-CU fp32-MFMA kernel (L2O_NO_PAIR=1) and the one-CU kernel with LDS-resident fragments (L2O_ONE_LDS=1)
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-36s value=%.4g G  ms/unroll=%.4f  fx_T=%r' % ('$LBL', d['value']/1e9, d['ms_per_unroll'], d['final_loss_fx_T']))"; }
for rep in 1 2; do
LBL="c4 pair-chunks" run --config 4 --steps 10
LBL="c4 one-CU LDS frags (L2O_ONE_LDS=1)" L2O_ONE_LDS=1 run --config 4 --steps 10
LBL="c2 B=256 pair-chunks" run --batch 256 --steps 10
LBL="c2 B=256 one-CU LDS frags" L2O_ONE_LDS=1 run --batch 256 --steps 10
LBL="c2 B=1024 pair-chunks" run --batch 1024 --steps 5 --unrolls-per-step 4
LBL="c2 B=1024 one-CU LDS frags" L2O_ONE_LDS=1 run --batch 1024 --steps 5 --unrolls-per-step 4
LBL="c2 B=128 one-CU LDS frags (=2)" L2O_ONE_LDS=2 run --steps 10
done
