#!/bin/bash
# End-of-round evidence on one box, one call: GPU suite, PMC traffic of the product build, rocprofv3 kernel trace of the bench, the bench itself, the replay.
# Usage (GPU box): bash tools/final_pass.sh <tag>   ->  gpurun_out/<tag>/...
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests -m gpu -q 2>&1 | tail -6 > "$OUT/gpu_tests.txt"
bash tools/pmc_traffic.sh "$TAG" > "$OUT/traffic.log" 2>&1
cp "gpurun_out/traffic_$TAG/hbm_traffic.json" profiles/r06_hbm_traffic.json
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -- python "$ROOT/bench.py" --no-extras > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err")
db=$(find "$OUT/prof" -name '*.db' | head -1)
[ -n "$db" ] && python tools/prof_summary.py "$db" > "$OUT/kernel_stats.txt" 2>&1
find "$OUT/prof" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/prof"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
for i in 1 2; do python tools/replay_bench.py 2>&1 | tail -1 >> "$OUT/replay.jsonl"; done
python tools/replay_bench.py --docs 4096 2>&1 | tail -1 >> "$OUT/replay.jsonl"
python tools/replay_bench.py --ops 2048 2>&1 | tail -1 >> "$OUT/replay.jsonl"
tail -3 "$OUT/gpu_tests.txt"; tail -c 1500 "$OUT/bench.json"; cat "$OUT/replay.jsonl" | cut -c1-300
