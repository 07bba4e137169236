#!/bin/bash
TAG=${1:-r03l}
O=$PWD/gpurun_out/$TAG; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for n in 4 16; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_i$n -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --instances $n > $O/bench_i$n.json 2> $O/bench_i$n.err
  db=$(ls $O/trace_i$n/*.db $O/trace_i$n/*/*.db 2>/dev/null | head -1)
  python $R/scripts/rocprof_summary.py $db $O/kernel_trace_i$n.txt > /dev/null
  python - <<PY
import sqlite3, sys
con = sqlite3.connect("$db")
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t == 'kernels'] or [t for t in tabs if 'kernel_dispatch' in t]
print("instances $n: views", view[:3])
try:
    rows = list(cur.execute("select name, start, end from kernels order by start"))
except Exception as e:
    print("no kernels view:", e); rows = []
rows = [r for r in rows if 'k_unroll_pair' in r[0] or 'k_combine' in r[0]]
# the last 320 unrolls = the timed region + the replays after it; look at the middle of the run
import statistics
pair = [(s, e) for n_, s, e in rows if 'k_unroll_pair' in n_]
comb = [(s, e) for n_, s, e in rows if 'k_combine' in n_]
mid = pair[100:400]
durs = [e - s for s, e in mid]
gaps = [mid[i + 1][0] - mid[i][1] for i in range(len(mid) - 1)]
print("instances $n: k_unroll_pair dur median %.1f us (min %.1f max %.1f); start-to-start median %.1f us; idle between pair kernels median %.1f us" % (
    statistics.median(durs) / 1e3, min(durs) / 1e3, max(durs) / 1e3,
    statistics.median([mid[i + 1][0] - mid[i][0] for i in range(len(mid) - 1)]) / 1e3, statistics.median(gaps) / 1e3))
PY
  rm -rf $O/trace_i$n
done 2>&1 | tee $O/summary.txt
