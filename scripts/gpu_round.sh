#!/bin/bash
# One GPU-box round: parity tests, smoke, bench (C2 + C3 + C4 shard), rocprof kernel trace
# and the HBM-traffic PMC passes.  Run via gpurun:  gpurun --timeout 1500 -- bash scripts/gpu_round.sh TAG
TAG=${1:-rXX}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tee $O/pytest_gpu.log | tail -4
grep -E "rel fx|vs C oracle" $O/pytest_gpu.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 3 2>$O/bench.err | tee $O/bench_c2.json | cut -c1-400
python bench.py --steps 5 --warmup 3 --problem lasso --net rnnprop --dims 512 --rows 256 --batch 256 --unroll 200 2>>$O/bench.err | tee $O/bench_c3.json | cut -c1-300
python bench.py --steps 10 --warmup 2 --problem rastrigin --net dm --dims 100 --batch 128 --unroll 100 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c4shard.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_trace -o trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_trace_bench.json 2>$O/prof_trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -o fetch -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_fetch_bench.json 2>$O/prof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -o write -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_write_bench.json 2>$O/prof_write.err
ls $O/prof_trace $O/prof_fetch $O/prof_write 2>&1 | head -20
cd $R
python bench.py --steps 5 --warmup 3 --problem lasso --net rnnprop --dims 512 --rows 256 --batch 256 --unroll 200 --shared-matrix 2>>$O/bench.err | tee $O/bench_c3_shared.json | cut -c1-200
python bench.py --steps 5 --warmup 3 --problem mnist --net rnnprop --unroll 100 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_mnist.json | cut -c1-200
python scripts/microbench/train_step_timing.py 2>/dev/null | tail -1 | tee $O/train_step.txt
python scripts/microbench/train_step_timing.py 128 128 100 2>/dev/null | tail -1 | tee -a $O/train_step.txt
python scripts/microbench/train_step_timing_mnist.py 2>/dev/null | tail -1 | tee -a $O/train_step.txt
# kernel traces of the training step (forward with history + BPTT + A^T Bm) and of the MLP-optimizee unroll
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_train_c2 -o t -- python $R/scripts/microbench/train_step_timing.py 128 128 100 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_train_mnist -o t -- python $R/scripts/microbench/train_step_timing_mnist.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_mnist -o t -- python $R/bench.py --steps 5 --warmup 3 --problem mnist --net rnnprop --unroll 100 --no-cpu-baseline > /dev/null 2>&1
cd $R
