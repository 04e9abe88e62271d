#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
: > $O/t12.log
for i in 1 2 3 4 5 6; do
  timeout 600 python3 -X faulthandler -m pytest tests/test_gpu_deferred.py -q -x > $O/t12_$i.log 2>&1; echo "run $i rc=$? $(tail -1 $O/t12_$i.log | cut -c1-80)" >> $O/t12.log
done
cat $O/t12.log
timeout 1500 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_faults.py > $O/t12_full.log 2>&1; echo "full rc=$?" >> $O/t12_full.log; tail -5 $O/t12_full.log
