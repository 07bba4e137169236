#!/bin/bash
# kernel trace of the T = 100 training step
TAG=${1:-r03t}
O=gpurun_out/$TAG; mkdir -p $O
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_train -o t -- python $R/scripts/microbench/train_step_timing.py 128 128 100 > $R/$O/trace_train.log 2>&1
cd $R
db=$(ls $O/trace_train/*/*_results.db $O/trace_train/*_results.db 2>/dev/null | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py $db $O/kernel_trace_train_T100.txt && head -12 $O/kernel_trace_train_T100.txt
rm -rf $O/trace_train
