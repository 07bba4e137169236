#!/usr/bin/env python
"""rocprofv3 --pmc passes (rocpd sqlite databases) of ONE bench.py command -> the small JSON that
bench.py's roofline block reads (profiles/rNN_counters_<tag>.json) + a text summary next to it.

    python scripts/counters_to_json.py OUT.json KERNEL_SUBSTRING '["quadratic","dm",128,128,100]' db1 [db2 ...]

Per counter: the per-launch TOTAL of the dominant kernel (the one whose name contains KERNEL_SUBSTRING), i.e. the
sum over the counter's hardware instances (rocprofv3 reports one row per shader engine for SQ_*: 32 rows per
dispatch on MI355X; one per XCD for GRBM_*), averaged over the kernel's dispatches.  GRBM_GUI_ACTIVE is kept as
the per-instance average (cycles the XCD was active).  FETCH_SIZE / WRITE_SIZE are stored as reported (KiB;
bench.py doubles FETCH_SIZE, MI355X_MICROARCH.md "HBM").  clock_hz_profiled = GRBM_GUI_ACTIVE / kernel duration
when that counter was collected."""
import glob
import json
import os
import sqlite3
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(out, kernel, workload, dbs):
    per, meta = {}, {"dbs": []}
    dur = {}
    for pat in dbs:
        for db in sorted(glob.glob(pat)):
            con = sqlite3.connect(db)
            cur = con.cursor()
            try:
                rows = list(cur.execute(
                    "select name, counter_name, count(distinct dispatch_id), sum(counter_value), count(*) from pmc_events "
                    "where name like ? group by name, counter_name", ("%" + kernel + "%",)))
            except Exception as e:
                print("skip %s: %s" % (db, e))
                continue
            for name, cname, n, total, nrows in rows:
                key = cname + ("_KiB" if cname in ("FETCH_SIZE", "WRITE_SIZE") else "")
                per[key] = total / nrows if cname.startswith("GRBM_") else total / n
                meta["kernel"] = name
                meta.setdefault("dispatches", {})[cname] = n
                meta.setdefault("instances", {})[cname] = nrows // n
            try:
                r = list(cur.execute("select name, count(*), avg(duration), vgpr_count, accum_vgpr_count, sgpr_count, "
                                     "workgroup_x, grid_x from kernels where name like ? group by name", ("%" + kernel + "%",)))
                if r:
                    _, n, avg_ns, vg, ag, sg, wgx, gx = r[0]
                    dur[os.path.basename(os.path.dirname(db))] = avg_ns
                    meta.update(vgpr=vg, agpr=ag, sgpr=sg, workgroup_x=wgx, grid_x=gx)
            except Exception as e:
                print("no kernels view in %s: %s" % (db, e))
            meta["dbs"].append(os.path.relpath(db))
    # which build these counters belong to (bench.py matches it against the library it times) and when they were taken
    try:
        from open_l2o_amd import _abi
        build_id = _abi.build_id()
    except Exception as e:
        print("no build id: %s" % e)
        build_id = None
    res = {"workload": json.loads(workload), "kernel": meta.get("kernel", kernel), "per_launch": per,
           "build_id": build_id, "collected_unix": time.time(),
           "kernel_ns_profiled": dur, "launch": {k: meta.get(k) for k in ("vgpr", "agpr", "sgpr", "workgroup_x", "grid_x")},
           "dispatches": meta.get("dispatches", {}), "instances_per_dispatch": meta.get("instances", {}),
           "source": "rocprofv3 --kernel-trace --pmc <group> -- python bench.py ... (one pass per counter group; "
                     "scripts/gpu_counters.sh), averaged over the timed + warm-up launches of the kernel"}
    # one wave per SIMD when the kernel's register budget exceeds 256 per lane
    regs = (meta.get("vgpr") or 0) + (meta.get("agpr") or 0)
    res["one_wave_per_simd"] = bool(regs > 256)
    if "GRBM_GUI_ACTIVE" in per and dur:
        ns = [v for k, v in dur.items()]
        res["clock_hz_profiled"] = per["GRBM_GUI_ACTIVE"] / (sum(ns) / len(ns) * 1e-9)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:])
