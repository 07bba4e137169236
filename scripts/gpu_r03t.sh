#!/bin/bash
# kernel trace of a training step: gpu_r03t.sh TAG [B D T KIND]
TAG=${1:-r03t}
shift
ARGS=${@:-128 128 100}
O=gpurun_out/$TAG; mkdir -p $O
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_train -o t -- python $R/scripts/microbench/train_step_host_phases.py $ARGS > $R/$O/trace_train.log 2>&1
cd $R
db=$(ls $O/trace_train/*/*_results.db $O/trace_train/*_results.db 2>/dev/null | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py $db $O/kernel_trace_train.txt && head -16 $O/kernel_trace_train.txt
rm -rf $O/trace_train
