#!/bin/bash
# last lease of the round: the whole GPU suite + the bench lines + training timings on the final build
TAG=${1:-r03last}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
(timeout 2400 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -12) | tee $O/pytest.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) | tee $O/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tee $O/bench_c2.json | cut -c1-260
timeout 300 python bench.py --config 3 --steps 5 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c3.json | cut -c1-200
timeout 300 python bench.py --config 4 --steps 10 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c4_one_gpu.json | cut -c1-200
timeout 300 python bench.py --config 5 --steps 5 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c5.json | cut -c1-200
(python scripts/microbench/train_epoch_timing.py 128 128 20 5 40 | head -2; python scripts/microbench/train_step_host_phases.py 128 128 100 | head -2; python scripts/microbench/train_step_host_phases.py 1024 100 20 rastrigin | head -2; python scripts/microbench/train_step_host_phases_mnist.py 20 64 | head -2) 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/train.txt
