#!/bin/bash
# One rocprofv3 PMC pass over the merge kernel: tools/pmc_pass.sh <tag> "<counters>" [phase_profile args]
set -u
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcpass_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc $CTRS --kernel-trace -d "$OUT/raw" -- python "$ROOT/tools/phase_profile.py" --no-phases --iters 2 --docs 8192 "$@" > "$OUT/run.log" 2>&1
db=$(find "$OUT/raw" -name '*.db' | head -1)
[ -n "$db" ] && python "$ROOT/tools/prof_summary.py" "$db" --pmc | grep per_dispatch | tee -a "$OUT/summary.txt"
rm -rf "$OUT/raw"
