#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprof kernel trace.  Run via gpurun.
set -x
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tee gpurun_out/pytest_gpu.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 20 --warmup 3 2>gpurun_out/bench.err | tee gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2>$GRAFT_REPO_ROOT/gpurun_out/prof.err
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof | head -30
