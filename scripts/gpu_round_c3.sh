#!/bin/bash
# Config-3 evidence round (streaming fused unroll): bench (with the CPU baseline), shared-A variant,
# rocprofv3 kernel trace and the two HBM-traffic PMC passes.  gpurun --timeout 600 -- bash scripts/gpu_round_c3.sh TAG
TAG=${1:-rXX}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
C3="--problem lasso --net rnnprop --dims 512 --rows 256 --batch 256 --unroll 200"
python bench.py --steps 5 --warmup 3 $C3 2>$O/bench.err | tee $O/bench_c3.json | cut -c1-300
python bench.py --steps 5 --warmup 3 $C3 --shared-matrix 2>>$O/bench.err | tee $O/bench_c3_shared.json | cut -c1-200
L2O_NO_UNROLL_CU=1 python bench.py --steps 5 --warmup 3 $C3 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c3_step_path.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_trace -o trace -- python $R/bench.py --steps 5 --warmup 3 $C3 --no-cpu-baseline > $O/prof_trace_bench.json 2>$O/prof_trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -o fetch -- python $R/bench.py --steps 5 --warmup 3 $C3 --no-cpu-baseline > $O/prof_fetch_bench.json 2>$O/prof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -o write -- python $R/bench.py --steps 5 --warmup 3 $C3 --no-cpu-baseline > $O/prof_write_bench.json 2>$O/prof_write.err
cd $R
for k in trace fetch write; do
  db=$(find $O/prof_$k -name "*.db" | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py $db $O/summary_$k.txt > /dev/null
done
head -12 $O/summary_trace.txt
grep -E "k_unroll_cu" $O/summary_fetch.txt $O/summary_write.txt | tail -4
