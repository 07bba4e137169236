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
            "select k.name, p.name, count(*), avg(e.value), sum(e.value) from rocpd_pmc_event e "
            "join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id "
            "join rocpd_info_kernel_symbol k on d.kernel_id = k.id group by k.name, p.name"))
        if rows:
            lines += ["", "## PMC counters (avg per dispatch)"]
            for kname, pname, n, avg, tot in rows:
                lines.append("%-48s %-24s n=%-5d avg=%.6g" % (kname[:48], pname, n, avg))
    except Exception as e:  # pragma: no cover
        lines += ["", "(no PMC tables: %s)" % e]
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
