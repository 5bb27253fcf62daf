#!/bin/bash
# Engine clock and power while the merge kernel runs back to back (rocm-smi polled beside bench.py's sustained leg): tools/clock_probe.sh <tag>
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
rocm-smi --showclocks --showpower > "$OUT/idle.txt" 2>&1
python bench.py --no-extras --no-cpu --sustain-s 12 > "$OUT/bench.json" 2> "$OUT/bench.err" &
BP=$!
for i in $(seq 1 60); do
  echo "--- t=$i" >> "$OUT/poll.txt"
  rocm-smi --showclocks --showpower 2>&1 | grep -E 'sclk|mclk|fclk|Power|power' >> "$OUT/poll.txt"
  sleep 0.5
  kill -0 $BP 2>/dev/null || break
done
wait $BP
grep -E 'sclk' "$OUT/poll.txt" | sort | uniq -c | sort -rn | head -12
grep -iE 'power' "$OUT/poll.txt" | awk '{print $NF}' | sort -n | uniq -c | tail -8
grep -E 'sclk|Power' "$OUT/idle.txt" | head -4
