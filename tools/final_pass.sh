#!/bin/bash
# End-of-round evidence on one box, one call: GPU suite, PMC traffic of the product build, rocprofv3 kernel trace of the bench, the bench itself, the replay
# (timings + instruction counters), per-phase instruction counters of the three lean builds, same-box A/B against the round before's library where it stands
# beside the product (peritext_amd/lib/exp_r5base.so).
# Usage (GPU box): bash tools/final_pass.sh <tag>   ->  gpurun_out/<tag>/...
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 > "$OUT/gpu_tests.txt"
bash tools/pmc_traffic.sh "$TAG" > "$OUT/traffic.log" 2>&1
cp "gpurun_out/traffic_$TAG/hbm_traffic.json" profiles/r06_hbm_traffic.json
cp "gpurun_out/traffic_$TAG/hbm_traffic.json" "$OUT/hbm_traffic.json"
for f in fetch write size; do cp "gpurun_out/traffic_$TAG/$f.txt" "$OUT/pmc_$f.txt" 2>/dev/null; done
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -- python "$ROOT/bench.py" --no-extras > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err")
db=$(find "$OUT/prof" -name '*.db' | head -1)
[ -n "$db" ] && python tools/prof_summary.py "$db" > "$OUT/kernel_stats.txt" 2>&1
find "$OUT/prof" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/prof"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
for i in 1 2; do python tools/replay_bench.py 2>&1 | tail -1 >> "$OUT/replay.jsonl"; done
python tools/replay_bench.py --docs 4096 2>&1 | tail -1 >> "$OUT/replay.jsonl"
python tools/replay_bench.py --ops 2048 2>&1 | tail -1 >> "$OUT/replay.jsonl"
REPLAY_ARGS="--docs 4096" bash tools/pmc_replay.sh "$TAG" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" > "$OUT/replay_insts.txt" 2>&1
if [ -f peritext_amd/lib/exp_diag.so ]; then
  bash tools/phase_insts.sh ${TAG}_config4 config4 65536 > "$OUT/insts_config4.txt" 2>&1
  bash tools/phase_insts.sh ${TAG}_config2 config2 524288 > "$OUT/insts_config2.txt" 2>&1
  bash tools/phase_insts.sh ${TAG}_config3 config3 196608 > "$OUT/insts_config3.txt" 2>&1
fi
if [ -f peritext_amd/lib/exp_r5base.so ]; then
  REPS=3 bash tools/r6_ab.sh ${TAG}_ab "config4 65536;config2 524288;config3 196608" peritext_amd/lib/libperitext_hip.so > "$OUT/ab_vs_round5.txt" 2>&1
  bash tools/r6_pmc_ab.sh ${TAG} peritext_amd/lib/exp_r5base.so peritext_amd/lib/libperitext_hip.so > "$OUT/counters_vs_round5.txt" 2>&1
fi
tail -3 "$OUT/gpu_tests.txt"; tail -c 1500 "$OUT/bench.json"; cat "$OUT/replay.jsonl" | cut -c1-300; cat "$OUT/ab_vs_round5.txt" 2>/dev/null
