#!/bin/bash
# rocprofv3 PMC passes over any command of this repo: tools/pmc_cmd.sh <tag> <kernel name prefix> "<python script + args>" "<counters pass 1>" ["<counters pass 2>" ...]
set -u
TAG=$1; KERN=$2; CMD=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmccmd_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
k=0
for CTRS in "$@"; do
  k=$((k+1))
  timeout 300 rocprofv3 --pmc $CTRS --kernel-trace -d "$OUT/raw$k" -- python $ROOT/$CMD > "$OUT/run$k.log" 2>&1
  db=$(find "$OUT/raw$k" -name '*.db' | head -1)
  [ -n "$db" ] && python "$ROOT/tools/prof_summary.py" "$db" --pmc --all | grep -E "$KERN.*(per_dispatch|grid=)" | tee -a "$OUT/summary.txt"
  rm -rf "$OUT/raw$k"
done
