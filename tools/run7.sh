#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
x="--no-cpu-baseline --no-fp32-exact"
rocprofv3 --kernel-trace --stats -d $O/prof_f32 -o f32 -- python3 bench.py --precision f32 $x --steps 20 > $O/bench_f32_under_rocprof.json 2>/dev/null
python3 tools/rocpd_stats.py $O/prof_f32/f32_results.db > $O/kernel_stats_f32.csv
python3 tools/step_timeline.py $O/prof_f32/f32_results.db 0 > $O/timeline_f32.txt
rm -rf $O/prof_f32
head -30 $O/kernel_stats_f32.csv | cut -c1-200
head -5 $O/timeline_f32.txt
