#!/usr/bin/env python3
"""Instruction counts per phase in ONE rocprofv3 PMC pass: the diagnostic kernel of a -DPTX_DIAG build (peritext_amd/lib/exp_diag.so) is launched once per phase,
truncated after that phase (ptx_diag_stop_after), and once whole; the differences between consecutive dispatches are the phases' own instructions.
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d OUT -- \\
        python tools/phase_insts.py --config config2 --docs 65536          (GPU box)
    python tools/phase_insts.py --table OUT/**/*_results.db --logs 65536      (anywhere: the per-phase table)"""
import argparse
import ctypes
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STOPS = [1, 11, 2, 12, 3, 4, 14, 16, 17, 18, 5, 6, 13, 7, 8, 0]
NAMES = {1: "P0 admission", 11: "P1 row loop", 2: "P1 tail", 12: "P3a", 3: "P3b", 4: "P3c", 14: "P3d after()", 16: "P3d successors", 17: "P3d jumping", 18: "P3d positions",
         5: "unpark", 6: "P4", 13: "P5a values", 7: "P5a marks", 8: "P5c", 0: "P5b + P6"}


def table(db, logs):
    c = sqlite3.connect(db)
    cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
    namecol = "kernel_name" if "kernel_name" in cols else cols[0]
    rows = list(c.execute("select dispatch_id, counter_name, sum(value) from counters_collection where %s like 'ptx_merge_kernel_diag%%' group by dispatch_id, counter_name order by dispatch_id" % namecol))
    ids = sorted({r[0] for r in rows})
    ids = ids[-len(STOPS):]  # (the warm-up launch comes first)
    per = {i: {} for i in ids}
    for d, n, v in rows:
        if d in per:
            per[d][n] = v
    names = sorted({r[1] for r in rows})
    print("%-16s" % "phase" + "".join("%22s" % n for n in names) + "   (per log)")
    prev = {n: 0.0 for n in names}
    for k, d in zip(STOPS, ids):
        print("%-16s" % NAMES[k] + "".join("%22.1f" % ((per[d].get(n, 0.0) - prev[n]) / logs) for n in names))
        prev = {n: per[d].get(n, 0.0) for n in names}
    print("%-16s" % "whole log" + "".join("%22.1f" % (prev[n] / logs) for n in names))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config2")
    ap.add_argument("--docs", type=int, default=65536)
    ap.add_argument("--lib", default="peritext_amd/lib/exp_diag.so")
    ap.add_argument("--table", default=None)
    ap.add_argument("--logs", type=int, default=65536)
    args = ap.parse_args()
    if args.table:
        return table(args.table, args.logs)
    from peritext_amd import abi, workloads
    from peritext_amd.engine import Engine

    c = workloads.gen_config(args.config)
    eng = Engine(0, flags=abi.FLAG_NO_ELEM_RANK, lib_path=os.path.join(ROOT, args.lib))
    eng.lib.ptx_diag_stop_after.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    db, _ = eng.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], args.docs, 2024, list_cap=2048)
    dr = eng.alloc_result(db)
    assert eng.lib.ptx_diag_stop_after(eng.ctx, 8) == 0
    eng.merge(db, dr)  # warm-up (diagnostic kernel: a stop is set)
    eng.sync()
    for k in STOPS:
        # 0 = run everything, but the diagnostic kernel is only chosen while a stop or the clocks are set: the last stamp index stands for "whole"
        assert eng.lib.ptx_diag_stop_after(eng.ctx, k if k else 10) == 0
        eng.merge(db, dr)
        eng.sync()
    print("logs", eng.n_logs(db))
    eng.free_result(dr)
    eng.free_batch(db)
    eng.close()


if __name__ == "__main__":
    main()
