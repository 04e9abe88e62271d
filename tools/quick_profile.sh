#!/bin/bash
# Quick look on the GPU box: bench lines of C2 / C3 and the full one-step timelines (every kernel, no duration floor).
#   gpurun --timeout 900 -- 'bash tools/quick_profile.sh TAG [configs...]'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
TAG=${1:-q}; shift
CFGS=${@:-c2 c3}
O=gpurun_out/$TAG; mkdir -p $O
x="--no-cpu-baseline --no-fp32-exact"
for n in $CFGS; do
  case $n in
    c2) a="";; c3) a="--rec local";; c4) a="--rec local --batch 32 --frames 40 --feat 2048";; c5) a="--rec local --batch 64 --frames 28 --feat 3584";;
  esac
  python3 bench.py $a $x $BENCH_EXTRA > $O/bench_$n.json 2> $O/bench_$n.err
  rocprofv3 --kernel-trace --stats -d $O/prof_$n -o $n -- python3 bench.py $a $x $BENCH_EXTRA --steps 30 > $O/bench_under_rocprof_$n.json 2>/dev/null
  python3 tools/rocpd_stats.py $O/prof_$n/${n}_results.db > $O/kernel_stats_$n.csv
  python3 tools/step_timeline.py $O/prof_$n/${n}_results.db 0 > $O/timeline_$n.txt
  rm -rf $O/prof_$n
done
ls -la $O
for n in $CFGS; do python3 -c "
import json; d=json.load(open('$O/bench_$n.json')); print('$n', d['ms_per_step'], d.get('roofline',{}).get('chain_launch_us'))"; done
