#!/bin/bash
# rocprofv3 kernel trace of meta-training steps (recording forward + BPTT + weight-gradient contraction + meta-step):
#   bash scripts/trace_train_step.sh OUTDIR [B D T [KIND]]      -> OUTDIR/kernel_trace_train_T<T>.txt
O=${1:?outdir}; B=${2:-128}; D=${3:-128}; T=${4:-100}; KIND=${5:-quadratic}
R=$PWD; mkdir -p $O; O=$(cd $O && pwd)
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_train -o t -- \
   python $R/scripts/microbench/train_step_timing.py $B $D $T $KIND > $O/trace_train.out 2> $O/trace_train.err)
db=$(ls $O/trace_train/*.db $O/trace_train/*/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py $db $O/kernel_trace_train_T$T.txt | head -14
rm -rf $O/trace_train
tail -2 $O/trace_train.out
