#!/bin/bash
# Repeats short bench runs with the Python fault handler on (and, every other run, glibc's checking allocator) to catch the
# rare host-side abort seen in round 4 ("corrupted size vs. prev_size"): prints rc per run, keeps stderr of failing runs.
O=gpurun_out/hunt; mkdir -p $O
x="--no-cpu-baseline --no-fp32-exact"
n=0
for rep in $(seq 1 ${REPS:-8}); do
  for args in "" "--batch 200" "--rec local"; do
    n=$((n+1))
    if [ $((n % 2)) = 0 ]; then pre="env MALLOC_CHECK_=3 LD_PRELOAD=libc_malloc_debug.so.0"; else pre="env"; fi
    $pre timeout 300 python3 -X faulthandler bench.py $args $x > $O/out_$n.txt 2> $O/err_$n.txt
    rc=$?
    echo "run $n [$args] check=$((1 - n % 2)) rc=$rc $(tail -c 120 $O/out_$n.txt | grep -o '"ms_per_step": [0-9.]*')"
    if [ $rc != 0 ]; then echo "---- stderr of run $n"; tail -60 $O/err_$n.txt; else rm -f $O/err_$n.txt $O/out_$n.txt; fi
  done
done
