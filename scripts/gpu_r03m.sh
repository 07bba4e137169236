#!/bin/bash
TAG=${1:-r03m}
O=gpurun_out/$TAG; mkdir -p $O
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
python scripts/microbench/train_step_host_phases_mnist.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/host_phases_mnist.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o t -- python $R/scripts/microbench/train_step_host_phases_mnist.py > $R/$O/trace.log 2>&1
cd $R
db=$(ls $O/trace/*/*_results.db $O/trace/*_results.db 2>/dev/null | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py $db $O/kernel_trace_train_mnist.txt && head -22 $O/kernel_trace_train_mnist.txt
rm -rf $O/trace
