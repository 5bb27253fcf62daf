#!/bin/bash
# replay throughput of several builds, a process each: tools/r6_replay_ab.sh <lib> [<lib> ...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"
for rep in 1 2; do for L in "$@"; do for D in 2048 4096; do python tools/replay_bench.py --lib $L --docs $D 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s logs %5d kernel_ms %.3f Gops/s %.3f launches %d'%('$L'.split('/')[-1], d['logs'], d['kernel_ms'], d['ops_per_s']/1e9, d['launches']))"; done; done; done
