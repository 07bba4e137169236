#!/bin/bash
# Round 3, lease 4: the whole GPU suite on the ABI-v9 build (caller-owned options, two-pass default, exact-gates option,
# co-residency probe), drift per kernel form, counters + bench of the new default kernel.
TAG=${1:-r03e}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
(timeout 1800 python -m pytest tests -q -m gpu --durations=12 -x 2>&1 | tail -30) | tee $O/pytest.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) | tee $O/smoke.log
timeout 600 python scripts/drift_forms.py > $O/drift_forms.txt 2>&1; grep -v amdgpu.ids $O/drift_forms.txt
bash scripts/gpu_counters.sh $PWD/$O c2 'k_unroll_pair<' '["quadratic","dm",128,128,100]' --steps 20 --warmup 3
bash scripts/gpu_counters.sh $PWD/$O c4 'k_unroll_pair<' '["rastrigin","dm",100,1024,100]' --config 4 --steps 10 --warmup 2
cp $O/counters_c2.json profiles/${TAG}_counters_c2.json; cp $O/counters_c4.json profiles/${TAG}_counters_c4.json
timeout 300 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tee $O/bench_c2.json | cut -c1-400
L2O_PAIR_NORMAL=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c2_normal.json | cut -c1-200
L2O_EXACT_GATES=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c2_exact.json | cut -c1-200
timeout 300 python bench.py --config 3 --steps 5 2>>$O/bench.err | tee $O/bench_c3.json | cut -c1-200
timeout 300 python bench.py --problem rastrigin --dims 100 --batch 128 --steps 10 2>>$O/bench.err | tee $O/bench_c4shard.json | cut -c1-200
timeout 300 python bench.py --config 4 --steps 10 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c4_one_gpu.json | cut -c1-200
timeout 300 python bench.py --config 5 --steps 5 2>>$O/bench.err | tee $O/bench_c5.json | cut -c1-200
tail -5 $O/bench.err
