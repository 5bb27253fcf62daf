#!/bin/bash
# L1-miss read requests (lines the vector L1 asks the L2 for) and L1 accesses per phase: the diagnostic kernel truncated after phase k, one PMC run each.
# Usage: tools/pmc_phases_mem.sh <tag> [phase_profile args]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcphm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in 1 11 2 12 3 4 14 5 6 13 7 8 0; do
  timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum --kernel-trace -d $OUT/k$k -- python $ROOT/tools/phase_profile.py --lib peritext_amd/lib/exp_diag.so --stop-after $k --no-phases --no-check --iters 2 "$@" > $OUT/k$k.log 2>&1
  db=$(find $OUT/k$k -name '*.db' | head -1)
  [ -n "$db" ] && python $ROOT/tools/prof_summary.py $db --pmc | grep per_dispatch | sed "s/^/stop_after=$k /" >> $OUT/summary.txt
  rm -rf $OUT/k$k
done
python3 - "$OUT/summary.txt" <<'PY'
import re, sys, collections
rows = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r"stop_after=(\d+)\s+(\S+)\s+(\S+)\s+dispatches=(\d+)\s+sum=(\S+)\s+per_dispatch=(\S+)", l)
    if m and m.group(2).startswith("ptx_merge_kernel"):
        rows.setdefault(int(m.group(1)), {})[m.group(3)] = float(m.group(6))
logs = 24576
names = {1: "P0 admission", 11: "P1 row loop", 2: "P1 tail", 12: "P3a", 3: "P3b", 4: "P3c", 14: "P3d tour", 5: "P3d rank+unpark", 6: "P4", 13: "P5a values", 7: "P5a marks", 8: "P5c", 0: "P5b+P6"}
prev = {}
print("%-18s %10s %10s %10s" % ("per log", "L1->L2 rd", "L1 access", "L1->L2 wr"))
for k in [1, 11, 2, 12, 3, 4, 14, 5, 6, 13, 7, 8, 0]:
    r = rows.get(k, {})
    d = {c: (r.get(c, 0) - prev.get(c, 0)) / logs for c in r}
    print("%-18s %10.0f %10.0f %10.0f" % (names[k], d.get("TCP_TCC_READ_REQ_sum", 0), d.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), d.get("TCP_TCC_WRITE_REQ_sum", 0)))
    prev = r
print("total", {c: round(v / logs) for c, v in rows.get(0, {}).items()})
PY
