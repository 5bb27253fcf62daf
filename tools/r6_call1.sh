#!/bin/bash
# round 6, GPU call 1: parity of the new build on the fixtures, same-box A/B against the round-5 library (+ knock-outs), per-phase instruction counters
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
OUT=gpurun_out/r6c1
mkdir -p $OUT
timeout 900 python tools/lib_ab.py --a peritext_amd/lib/exp_r5base.so --b peritext_amd/lib/libperitext_hip.so peritext_amd/lib/exp_kodigest.so --config config4 --docs 65536 --rounds 3 > $OUT/ab_config4.json 2> $OUT/ab_config4.err
tail -12 $OUT/ab_config4.err
for c in "config2 524288" "config3 196608" "config5 24576"; do set -- $c; timeout 600 python tools/lib_ab.py --a peritext_amd/lib/exp_r5base.so --b peritext_amd/lib/libperitext_hip.so --config $1 --docs $2 --no-parity --rounds 3 > $OUT/ab_$1.json 2> $OUT/ab_$1.err; tail -7 $OUT/ab_$1.err; done
bash tools/phase_insts.sh r6c1_config4 config4 65536 | tail -20
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $OUT/gpu_tests.txt; cat $OUT/gpu_tests.txt
