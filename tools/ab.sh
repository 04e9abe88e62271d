#!/bin/bash
# A/B of environment knobs on the GPU box: tools/ab.sh "bench args" "ENV1=.. ENV2=.." "ENV1=.." ...   (first column: ms per step, then the phase table)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
args="$1"; shift
for e in "$@"; do
  env $e python3 bench.py $args --no-cpu-baseline --no-fp32-exact 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d.get('phases') or {}
print('%-40s %.4f ms  loss %.5f | '%('$e', d['ms_per_step'], d['config']['loss']) + ' '.join('%s=%s'%(k.replace('_us','').replace('chain_','c:').replace('gap_','g:'),v) for k,v in p.items()))"
done
