#!/bin/bash
# lease: final-core measurements: long-horizon numbers of every form, counters of the default kernel, bench lines
TAG=${1:-r03z}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
(timeout 900 python -m pytest tests/test_trained_parity.py -q -m gpu -s 2>&1 | grep -E "c2 |c4shard |C3 trained|segment|passed|failed" | sed 's/^\.*//') > $O/trained_parity_numbers.txt; grep -E "T=|passed|failed" $O/trained_parity_numbers.txt | cut -c1-200
(timeout 2400 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -14) | tee $O/pytest.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) | tee $O/smoke.log
bash scripts/gpu_counters.sh $PWD/$O c2 'k_unroll_pair<' '["quadratic","dm",128,128,100]' --steps 20 --warmup 3
bash scripts/gpu_counters.sh $PWD/$O c4 'k_unroll_pair<' '["rastrigin","dm",100,1024,100]' --config 4 --steps 10 --warmup 2
cp $O/counters_c2.json profiles/${TAG}_counters_c2.json; cp $O/counters_c4.json profiles/${TAG}_counters_c4.json
timeout 300 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tee $O/bench_c2.json | cut -c1-300
L2O_PAIR_NORMAL=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c2_normal.json | cut -c1-200
L2O_EXACT_GATES=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c2_exact.json | cut -c1-200
timeout 300 python bench.py --config 3 --steps 5 2>>$O/bench.err | tee $O/bench_c3.json | cut -c1-200
timeout 300 python bench.py --config 4 --steps 10 2>>$O/bench.err | tee $O/bench_c4_one_gpu.json | cut -c1-200
timeout 300 python bench.py --config 5 --steps 5 2>>$O/bench.err | tee $O/bench_c5.json | cut -c1-200
timeout 300 python bench.py --batch 256 --steps 10 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c2_b256.json | cut -c1-200
L2O_BENCH_BACKEND=gloo L2O_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 2>$O/bench2.err | tee $O/bench_2ranks_one_device.json | cut -c1-200
python scripts/microbench/train_step_timing.py 2>/dev/null | tail -1 | tee $O/train_step.txt
python scripts/microbench/train_step_timing.py 128 128 100 2>/dev/null | tail -1 | tee -a $O/train_step.txt
