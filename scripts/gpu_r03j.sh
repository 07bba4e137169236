#!/bin/bash
TAG=${1:-r03j}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
(timeout 900 python -m pytest tests/test_meta_api.py tests/test_bench_multi_rank.py -q -m gpu 2>&1 | tail -8) | tee $O/pytest_meta.log
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c2 run: value %.4g replay %.4g ms/unroll %.4f host enqueue %.4f kernel %.4f' % (d['value'], d['value_replayed_problem'], d['ms_per_unroll'], d['host_enqueue_ms_per_unroll'], d['roofline']['kernel_ms_avg']))"; done | tee $O/bench_c2_host.txt
for c in 3 4 5; do timeout 300 python bench.py --config $c --steps 10 --no-cpu-baseline 2>>$O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config run: value %.4g replay %.4g ms/unroll %.4f host enqueue %.4f kernel %.4f' % (d['value'], d['value_replayed_problem'], d['ms_per_unroll'], d['host_enqueue_ms_per_unroll'], d['roofline']['kernel_ms_avg']))"; done | tee -a $O/bench_c2_host.txt
