#!/bin/bash
TAG=${1:-r03o}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8) | tee $O/pytest.log
for c in 4 2; do timeout 300 python bench.py --config $c --steps 10 --no-cpu-baseline 2>>$O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config %s: value %.4g replay %.4g ms/unroll %.4f host enqueue %.4f kernel %.4f fxT %.6g' % (sys.argv[1], d['value'], d['value_replayed_problem'], d['ms_per_unroll'], d['host_enqueue_ms_per_unroll'], d['roofline']['kernel_ms_avg'], d['final_loss_fx_T']))" $c; done | tee $O/bench_all.txt
timeout 300 python bench.py --problem rastrigin --dims 100 --batch 128 --steps 10 --no-cpu-baseline 2>>$O/bench.err | cut -c1-200
