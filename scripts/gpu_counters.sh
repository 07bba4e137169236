#!/bin/bash
# PMC passes of one bench.py command: bash scripts/gpu_counters.sh OUTDIR TAG KERNEL 'WORKLOAD_JSON' <bench args...>
# One rocprofv3 run per counter group (SQ has 8 slots, FETCH_SIZE / WRITE_SIZE do not share a pass); --pmc is
# never combined with the hip/hsa trace domains.  Writes OUTDIR/counters_TAG.json + the kernel-trace stats.
O=$1; TAG=$2; KERNEL=$3; WL=$4; shift 4
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_${TAG}_$name -o p -- python $R/bench.py --no-cpu-baseline --no-also "${BENCH_ARGS[@]}" > $O/pmc_${TAG}_$name.json 2> $O/pmc_${TAG}_$name.err || echo "pass $name failed (see $O/pmc_${TAG}_$name.err)"
}
BENCH_ARGS=("$@")
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$TAG -o t -- python $R/bench.py --no-cpu-baseline --no-also "${BENCH_ARGS[@]}" > $O/trace_$TAG.json 2> $O/trace_$TAG.err
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM
run sq3 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
# LDS-array occupancy and conflicts (the kernels with their gate-GEMM fragments in LDS: DESIGN 3.1d); L2O_COUNTERS_LDS=1
[ -n "$L2O_COUNTERS_LDS" ] && run lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
python scripts/counters_to_json.py $O/counters_$TAG.json "$KERNEL" "$WL" "$O/pmc_${TAG}_*/*.db" "$O/pmc_${TAG}_*/*/*.db" > $O/counters_$TAG.log 2>&1
for d in $O/trace_$TAG; do
  db=$(ls $d/*.db $d/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python scripts/rocprof_summary.py $db $O/kernel_trace_$TAG.txt > /dev/null
done
# the rocpd databases are tens of MB per pass (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf $O/pmc_${TAG}_* $O/trace_$TAG
tail -12 $O/counters_$TAG.log
