#!/bin/bash
# full GPU suite on the build with k_atb_bx3 + kernel trace of the T = 100 training step
TAG=${1:-r03r}
O=gpurun_out/$TAG; mkdir -p $O
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
(timeout 2400 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -14) | tee $O/pytest.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) | tee $O/smoke.log
(for a in "128 128 20" "128 128 100"; do timeout 300 python scripts/microbench/train_step_timing.py $a; done) 2>&1 | grep "train step" | tee $O/train_step.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_train -o t -- python $R/scripts/microbench/train_step_timing.py 128 128 100 > $R/$O/trace_train.log 2>&1
cd $R
db=$(ls $O/trace_train/*/*_results.db $O/trace_train/*_results.db 2>/dev/null | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py $db $O/kernel_trace_train_T100.txt && head -24 $O/kernel_trace_train_T100.txt
rm -rf $O/trace_train
