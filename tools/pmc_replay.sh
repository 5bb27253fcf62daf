#!/bin/bash
# rocprofv3 PMC passes over the patch-stream replay kernel: tools/pmc_replay.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcreplay_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
k=0
for CTRS in "$@"; do
  k=$((k+1))
  timeout 240 rocprofv3 --pmc $CTRS --kernel-trace -d "$OUT/raw$k" -- python "$ROOT/tools/replay_bench.py" ${REPLAY_ARGS:-} > "$OUT/run$k.log" 2>&1
  db=$(find "$OUT/raw$k" -name '*.db' | head -1)
  [ -n "$db" ] && python "$ROOT/tools/prof_summary.py" "$db" --pmc --all | grep -E 'ptx_replay.*(per_dispatch|grid=)' | tee -a "$OUT/summary.txt"
  rm -rf "$OUT/raw$k"
done
