#!/bin/bash
# Any PMC counters per phase (the diagnostic kernel truncated after phase k, one rocprofv3 run each; differences of consecutive k are the phase's own).
# Usage: tools/pmc_phases_any.sh <tag> "<counters>" [phase_profile args]    ->  gpurun_out/pmcpha_<tag>/{summary,table}.txt
set -u
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcpha_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in 1 11 2 12 3 4 14 5 6 13 7 8 0; do
  timeout 200 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/k$k -- python $ROOT/tools/phase_profile.py --lib peritext_amd/lib/exp_diag.so --stop-after $k --no-phases --no-check --iters 2 "$@" > $OUT/k$k.log 2>&1
  db=$(find $OUT/k$k -name '*.db' | head -1)
  [ -n "$db" ] && python $ROOT/tools/prof_summary.py $db --pmc | grep per_dispatch | sed "s/^/stop_after=$k /" >> $OUT/summary.txt
  rm -rf $OUT/k$k
done
python3 - "$OUT/summary.txt" "$CTRS" <<'PY' | tee $OUT/table.txt
import re, sys, collections
rows = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r"stop_after=(\d+)\s+(\S+)\s+(\S+)\s+dispatches=(\d+)\s+sum=(\S+)\s+per_dispatch=(\S+)", l)
    if m and m.group(2).startswith("ptx_merge_kernel"):
        rows.setdefault(int(m.group(1)), {})[m.group(3)] = float(m.group(6))
ctrs = sys.argv[2].split()
logs = 24576
names = {1: "P0 admission", 11: "P1 row loop", 2: "P1 tail", 12: "P3a", 3: "P3b", 4: "P3c", 14: "P3d tour", 5: "P3d rank+unpark", 6: "P4", 13: "P5a values", 7: "P5a marks", 8: "P5c", 0: "P5b+P6"}
prev = {}
print("%-18s " % "per log" + " ".join("%22s" % c for c in ctrs))
for k in [1, 11, 2, 12, 3, 4, 14, 5, 6, 13, 7, 8, 0]:
    r = rows.get(k, {})
    print("%-18s " % names[k] + " ".join("%22.0f" % ((r.get(c, 0) - prev.get(c, 0)) / logs) for c in ctrs))
    prev = r
print("%-18s " % "total" + " ".join("%22.0f" % (prev.get(c, 0) / logs) for c in ctrs))
PY
