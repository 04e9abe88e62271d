#!/bin/bash
# A/B of environment knobs on the GPU box, interleaved rounds (box-to-box and run-to-run spread is ~1 %):
#   ROUNDS=3 tools/ab.sh "bench args" "ENV1=.. ENV2=.." "ENV1=.." ...   -> per variant: ms per step of every round, the median, the last phase table
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
args="$1"; shift
R=${ROUNDS:-3}
rm -f /tmp/ab_*.txt
for r in $(seq $R); do
  i=0
  for e in "$@"; do
    env $e python3 bench.py $args --no-cpu-baseline --no-fp32-exact 2>/dev/null >> /tmp/ab_$i.txt
    i=$((i+1))
  done
done
i=0
for e in "$@"; do
  python3 -c "
import json,sys,statistics
rows=[json.loads(l) for l in open('/tmp/ab_$i.txt') if l.strip()]
ms=[d['ms_per_step'] for d in rows]; p=rows[-1].get('phases') or {}
print('%-44s median %.4f ms  (%s)  loss %.5f | '%('$e', statistics.median(ms), ' '.join('%.4f'%m for m in ms), rows[-1]['config']['loss']) + ' '.join('%s=%s'%(k.replace('_us','').replace('chain_','c:').replace('gap_','g:').replace('group_','G:'),v) for k,v in p.items() if not isinstance(v, dict)))"
  i=$((i+1))
done
