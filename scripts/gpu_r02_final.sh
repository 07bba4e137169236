#!/bin/bash
# full verification + the measurements behind the bench line: tests, smoke, bench C2/C3/C4shard/C5, 2-rank path, PMC passes
TAG=${1:-r02_final}
O=gpurun_out/$TAG; mkdir -p $O
(timeout 1200 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -16) | tee $O/pytest.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) | tee $O/smoke.log
bash scripts/gpu_counters.sh $PWD/$O c2 k_unroll_pairh '["quadratic","dm",128,128,100]' --steps 20 --warmup 3
bash scripts/gpu_counters.sh $PWD/$O c3 k_unroll_cu '["lasso","rnnprop",512,256,200,256]' --config 3 --steps 5 --warmup 2
cp $O/counters_c2.json profiles/${TAG}_counters_c2.json; cp $O/counters_c3.json profiles/${TAG}_counters_c3.json   # (so that the bench runs below see them)
timeout 300 python bench.py 2>$O/bench.err | tee $O/bench_c2.json | cut -c1-300
timeout 300 python bench.py --config 3 --steps 5 2>>$O/bench.err | tee $O/bench_c3.json | cut -c1-300
timeout 300 python bench.py --problem rastrigin --dims 100 --batch 128 --steps 10 2>>$O/bench.err | tee $O/bench_c4shard.json | cut -c1-300
timeout 300 python bench.py --config 4 --steps 10 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c4_one_gpu.json | cut -c1-300
timeout 300 python bench.py --config 5 --steps 5 2>>$O/bench.err | tee $O/bench_c5.json | cut -c1-300
L2O_BENCH_BACKEND=gloo L2O_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 2>$O/bench2.err | tee $O/bench_2ranks_one_device.json | cut -c1-300
python scripts/microbench/train_step_timing.py 2>/dev/null | tail -1 | tee $O/train_step.txt
python scripts/microbench/train_step_timing.py 128 128 100 2>/dev/null | tail -1 | tee -a $O/train_step.txt
python scripts/microbench/train_step_timing_mnist.py 2>/dev/null | tail -1 | tee -a $O/train_step.txt
