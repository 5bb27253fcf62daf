#!/bin/bash
# One PMC pass: tools/pmc_one.sh <tag> "<counters>" [phase_profile args]
set -u
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc1_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc $CTRS --kernel-trace -d "$OUT/raw" -- python "$ROOT/tools/phase_profile.py" --no-phases --iters 2 "$@" > "$OUT/run.log" 2>&1
db=$(find "$OUT/raw" -name '*.db' | head -1)
[ -n "$db" ] && python "$ROOT/tools/prof_summary.py" "$db" --pmc | grep per_dispatch > "$OUT/summary.txt" 2>&1
find "$OUT" -name '*.db' -delete
