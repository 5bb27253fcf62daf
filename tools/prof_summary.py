#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite .db, the default output of this ROCm) as a per-kernel stats
table: the same figures `rocprofv3 --stats` prints in its kernel_stats CSV.  Usage:
    python tools/prof_summary.py gpurun_out/prof_x/**/NNN_results.db [--pmc] > profiles/rNN_x.txt"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    print("# source: %s" % db)
    print("# per-kernel statistics (durations in ns)")
    print("%-60s %8s %14s %14s %14s %14s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    for name, n, s, a, mn, mx in rows:
        print("%-60s %8d %14d %14.0f %14d %14d %7.2f%%" % (name[:60], n, s, a, mn, mx, 100.0 * s / tot))
    print()
    print("# dispatches of the merge kernel (grid, workgroup, LDS, registers, duration)")
    q = "select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size, duration from kernels where name like 'ptx_%' order by start"
    for r in c.execute(q):
        print("%-40s grid=%-9d wg=%-5d lds=%-7d vgpr=%-4d agpr=%-4d sgpr=%-4d scratch=%-4d dur_ns=%d" % ((r[0][:40],) + r[1:]))
    if "--pmc" in sys.argv:
        print()
        print("# PMC counters (sum over dispatches per kernel)")
        try:
            cur = c.execute("select * from counters_collection limit 1")
            cols = [d[0] for d in cur.description]
            print("# columns: %s" % cols)
            namecol = "kernel_name" if "kernel_name" in cols else cols[0]
            like = "ptx_%" if "--all" in sys.argv else "ptx_merge%"
            q = "select %s, counter_name, count(distinct dispatch_id), sum(value) from counters_collection where %s like '%s' group by %s, counter_name" % (namecol, namecol, like, namecol)
            for r in c.execute(q):
                print("%-40s %-28s dispatches=%-5d sum=%.6g per_dispatch=%.6g" % (str(r[0])[:40], r[1], r[2], r[3], r[3] / max(r[2], 1)))
        except Exception as e:  # noqa: BLE001
            print("# no counters: %s" % e)


if __name__ == "__main__":
    main()
