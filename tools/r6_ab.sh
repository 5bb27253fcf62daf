#!/bin/bash
# A/B of several builds, each measured in a process of its own (same allocation sequence, same device addresses), the list gone through REPS times:
#   tools/r6_ab.sh <tag> "<configs: 'config4 65536;config2 524288'>" <lib> [<lib> ...]   (GPU box; BASE = the reference build, default exp_r5base.so)
set -u
TAG=$1; CFGS=$2; shift 2
LIBS=(${BASE:-peritext_amd/lib/exp_r5base.so} "$@")
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"; mkdir -p gpurun_out/$TAG
IFS=';' read -ra CS <<< "$CFGS"
for c in "${CS[@]}"; do
  CFG=$(echo $c | awk '{print $1}'); DOCS=$(echo $c | awk '{print $2}')
  : > gpurun_out/$TAG/ab_$CFG.jsonl
  for rep in $(seq 1 ${REPS:-2}); do for L in "${LIBS[@]}"; do timeout 300 python tools/r6_time.py --lib $L --config $CFG --docs $DOCS ${EXTRA:-} >> gpurun_out/$TAG/ab_$CFG.jsonl 2>> gpurun_out/$TAG/ab_$CFG.err; done; done
  python - gpurun_out/$TAG/ab_$CFG.jsonl $CFG <<'PY'
import json,sys
by={}; sha={}; order=[]
for l in open(sys.argv[1]):
    d=json.loads(l); b=d['build']
    if b not in by: order.append(b)
    by.setdefault(b,[]).extend(d['kernel_ms']); sha.setdefault(b,set()).add((d['results_sha1'],d['max_status']))
base=min(by[order[0]])
for b in order:
    v=by[b]; print(sys.argv[2], '%-24s'%b, ' '.join('%.3f'%x for x in v), 'min %.4f'%min(v), '%+.2f%%'%(100*(min(v)/base-1)), 'same_results' if sha[b]==sha[order[0]] else 'RESULTS DIFFER %s'%sha[b])
PY
done
