#!/bin/bash
# One configuration under rocprofv3: kernel statistics + the timeline of one replayed step.   tools/quick_profile.sh <name> "<bench args>"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
n=$1; a="$2"; O=gpurun_out/${RND:-r06}; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_$n -o $n -- python3 bench.py $a --no-cpu-baseline --no-fp32-exact --steps 30 > $O/bench_under_rocprof_$n.json 2>/dev/null
python3 tools/rocpd_stats.py $O/prof_$n/${n}_results.db > $O/kernel_stats_$n.csv
python3 tools/step_timeline.py $O/prof_$n/${n}_results.db 5 > $O/timeline_$n.txt
rm -rf $O/prof_$n
cat $O/timeline_$n.txt
