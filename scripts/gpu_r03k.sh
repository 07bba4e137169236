#!/bin/bash
TAG=${1:-r03k}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>>$O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s value %.4g replay %.4g ms/unroll %.4f host enqueue %.4f kernel %.4f' % (' '.join(sys.argv[1:]), d['value'], d['value_replayed_problem'], d['ms_per_unroll'], d['host_enqueue_ms_per_unroll'], d['roofline']['kernel_ms_avg']))" "$@"; }
for rep in 1 2; do
  run --instances 1
  run --instances 4
  run --instances 2
  run --instances 16
  L2O_EXACT_GATES=1 run --instances 4
  run --instances 4 --unrolls-per-step 4
done | tee $O/instances_ab.txt
