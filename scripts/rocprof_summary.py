#!/usr/bin/env python
"""Turn a rocprofv3 rocpd sqlite database (--kernel-trace --stats, optional --pmc) into the
small text summary that gets committed under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    lines = ["# rocprofv3 summary of %s" % db, "", "## kernel stats (top_kernels)",
             "%-72s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, total, avg, pct in cur.execute("select * from top_kernels"):
        lines.append("%-72s %8d %14.3f %12.3f %7.2f" % (name[:72], calls, total, avg, pct))
    lines += ["", "## per-kernel dispatch geometry / resources (kernels view)",
              "%-40s %6s %10s %10s %10s %6s %6s %6s %8s %8s %7s %8s" %
              ("kernel", "n", "avg_ns", "min_ns", "max_ns", "vgpr", "agpr", "sgpr", "lds", "scratch", "wg_x", "grid_x")]
    q = ("select name, count(*), avg(duration), min(duration), max(duration), vgpr_count, accum_vgpr_count, "
         "sgpr_count, lds_size, scratch_size, workgroup_x, grid_x from kernels group by name, grid_x order by 3 desc")
    for r in cur.execute(q):
        lines.append("%-40s %6d %10.0f %10d %10d %6s %6s %6s %8s %8s %7s %8s" % ((r[0][:40],) + tuple(r[1:])))
    try:
        rows = list(cur.execute(
            "select name, counter_name, count(*), avg(counter_value), min(counter_value), max(counter_value) "
            "from pmc_events group by name, counter_name order by 4 desc"))
        if rows:
            lines += ["", "## PMC counters per dispatch (pmc_events view; FETCH_SIZE / WRITE_SIZE are in KiB)",
                      "%-56s %-14s %6s %14s %14s %14s" % ("kernel", "counter", "n", "avg", "min", "max")]
            for kname, cname, n, avg, mn, mx in rows:
                lines.append("%-56s %-14s %6d %14.3f %14.3f %14.3f" % (kname[:56], cname, n, avg, mn, mx))
    except Exception as e:  # pragma: no cover
        lines += ["", "(no PMC tables: %s)" % e]
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
