#!/bin/bash
# same-box A/B of the product against peritext_amd/lib/exp_<name>.so on BASELINE configs #2, #3, #4 (tools/lib_ab.py): tools/ab3.sh <name> [extra lib_ab args]
NAME=$1; shift
for c in "config2 524288" "config3 196608" "config4 65536"; do set -- $c "${@:3}"; timeout 300 python tools/lib_ab.py --b peritext_amd/lib/exp_$NAME.so --config $1 --docs $2 --no-parity --rounds 3 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
by={}
for t in d['timing']: by.setdefault(t['build'],[]).append(t['kernel_ms'])
for b,v in by.items(): print('$1', b, ' '.join('%.4f'%x for x in v), 'min %.4f'%min(v))
print('identical', d.get('identical_results'))
"; done
