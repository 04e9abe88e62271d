#!/bin/bash
# A/B of environment knobs on the benchmark step: tools/ab_env.sh "<bench args>" VAR1=a,b VAR2=c,d ...   (all combinations, 2 repetitions)
args="$1"; shift
combos=("")
for spec in "$@"; do
  var=${spec%%=*}; vals=${spec#*=}
  new=()
  for c in "${combos[@]}"; do for v in ${vals//,/ }; do new+=("$c $var=$v"); done; done
  combos=("${new[@]}")
done
for rep in 1 2; do
  for c in "${combos[@]}"; do
    ms=$(env $c python3 bench.py --no-cpu-baseline --no-fp32-exact $args 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep $rep |$c | $ms ms"
  done
done
