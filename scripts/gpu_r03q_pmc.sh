#!/bin/bash
# PMC passes of the weight-gradient contraction microbenchmark (k_atb fp32 pipe vs k_atb_bx3): HBM traffic against the known
# byte count (4 (KA + KB) R = 1.59 GB per launch), LDS bank conflicts, matrix-pipe busy cycles.  --pmc passes carry
# --kernel-trace only.
TAG=${1:-r03q6}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$name -o p -- $R/scripts/microbench/atb_bx3_bench 1638400 1 > $O/pmc_$name.log 2>&1 || echo "pass $name failed"
  db=$(ls $O/pmc_$name/*.db $O/pmc_$name/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/scripts/rocprof_summary.py $db $O/atb_pmc_$name.txt > /dev/null
  rm -rf $O/pmc_$name
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM
cd $R
for f in $O/atb_pmc_*.txt; do echo "== $f"; sed -n '/PMC counters/,$p' $f | grep -v "k_atb_reduce" | head -24; done
