#!/bin/bash
# round 2, first GPU call: tests (ABI v6, soak, long horizons, C client), smoke, bench, SQ / HBM counter passes of C2 and C3
O=gpurun_out/r02a; mkdir -p $O
( rocprofv3 -L > $O/counters_avail.txt 2>&1 ; grep -c . $O/counters_avail.txt ) | tail -1
(timeout 1200 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -40) | tee $O/pytest.log
grep -E "soak|T=1000|C3 |abi_smoke" $O/pytest.log | head
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) | tee $O/smoke.log
timeout 300 python bench.py 2>$O/bench.err | tee $O/bench_c2.json | cut -c1-600
L2O_BENCH_BACKEND=gloo L2O_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 2>$O/bench2.err | tee $O/bench_2ranks_one_device.json | cut -c1-1200
bash scripts/gpu_counters.sh $PWD/$O c2 k_unroll_pair '["quadratic","dm",128,128,100]' --steps 20 --warmup 3
bash scripts/gpu_counters.sh $PWD/$O c3 k_unroll_cu '["lasso","rnnprop",512,256,200,256]' --config 3 --steps 5 --warmup 2
ls $O | head -50
