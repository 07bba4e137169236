#!/bin/bash
# fused-unroll throughput against the batch size (run on the GPU box)
p='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%s  %.3g G coord-steps/s  kernel %.4f ms" % (d["config"]["workload"][:70], d["value"]/1e9, d["roofline"]["kernel_ms_avg"]))'
for b in 128 256 512 1024; do
  python bench.py --steps 5 --warmup 2 --problem rastrigin --net dm --dims 100 --batch $b --unroll 100 --no-cpu-baseline 2>/dev/null | python -c "$p"
done
for b in 64 128 256 512; do
  python bench.py --steps 5 --warmup 2 --batch $b --no-cpu-baseline 2>/dev/null | python -c "$p"
done
