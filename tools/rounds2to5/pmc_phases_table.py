#!/usr/bin/env python3
"""Per-phase instruction table from tools/pmc_phases.sh output: python tools/pmc_phases_table.py gpurun_out/pmcph_X/summary.txt LOGS [LAUNCHES]
A merge is one launch of the kernel + possibly a second one for the few logs that need a larger LDS window: the counters of all
merge kernels of a run are summed and divided by the number of merges (tools/phase_profile.py --iters 2 makes 3)."""
import collections
import re
import sys

rows = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r"stop_after=(\d+)\s+(\S+)\s+(\S+)\s+dispatches=(\d+)\s+sum=(\S+)\s+per_dispatch=(\S+)", l)
    if m and m.group(2).startswith("ptx_merge_kernel"):
        d = rows.setdefault(int(m.group(1)), {})
        d[m.group(3)] = d.get(m.group(3), 0.0) + float(m.group(5))
logs = int(sys.argv[2]) if len(sys.argv) > 2 else 24576
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for d in rows.values():
    for c in d:
        d[c] /= launches
order = [1, 11, 2, 12, 3, 4, 14, 5, 6, 13, 7, 8, 0]
names = {1: "P0 admission", 11: "P1 row loop", 2: "P1 tail: census, dup, scan", 12: "P3a index+parents", 3: "P3b scatter+checks", 14: "P3d tour", 5: "P3d ranking+unpark", 13: "P5a values", 7: "P5a mark intervals", 99: "", 4: "P3c child order", 6: "P4 tombstones", 8: "P5c comments", 0: "P5b trees+spans"}
prev = {}
print("per log: %-22s %8s %8s %8s %8s %8s | %10s %10s %10s" % ("phase", "VALU", "SALU", "LDS", "VMEM", "BRANCH", "wavecyc(q)", "wait(q)", "active(q)"))
for k in order:
    r = rows.get(k, {})
    d = {c: (r.get(c, 0) - prev.get(c, 0)) / logs for c in r}
    g = lambda c: d.get(c, 0)  # noqa: E731
    print("         %-22s %8.0f %8.0f %8.0f %8.0f %8.0f | %10.0f %10.0f %10.0f" % (names[k], g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_BRANCH"), g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY"), g("SQ_ACTIVE_INST_ANY")))
    prev = r
t = rows.get(0, {})
print("total   ", {c: round(v / logs) for c, v in t.items()})
