#!/bin/bash
TAG=${1:-r03n}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6) | tee $O/pytest.log
for c in 2 3 4 5; do timeout 300 python bench.py --config $c --steps 10 --no-cpu-baseline 2>>$O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config %s: value %.4g replay %.4g ms/unroll %.4f host enqueue %.4f kernel %.4f' % (sys.argv[1], d['value'], d['value_replayed_problem'], d['ms_per_unroll'], d['host_enqueue_ms_per_unroll'], d['roofline']['kernel_ms_avg']))" $c; done | tee $O/bench_all.txt
