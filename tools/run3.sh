#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_deferred.py -v -x > $O/t3_deferred.log 2>&1; echo "rc=$?" >> $O/t3_deferred.log
grep -E "PASSED|FAILED|SKIPPED|rc=|Segmentation" $O/t3_deferred.log | tail -15
timeout 1200 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_faults.py --deselect tests/test_gpu_deferred.py > $O/t3_pytest.log 2>&1; echo "pytest rc=$?" >> $O/t3_pytest.log
tail -15 $O/t3_pytest.log
