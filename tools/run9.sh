#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
ROUNDS=5 bash tools/ab.sh "" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT=" 2>&1 | tee $O/ab_transposes.txt | cut -c1-120
ROUNDS=3 bash tools/ab.sh "--rec local" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT=" 2>&1 | tee -a $O/ab_transposes.txt | cut -c1-120
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_knobs.py tests/test_gpu_configs.py -q -x > $O/t9_pytest.log 2>&1; echo "pytest rc=$?" >> $O/t9_pytest.log
tail -4 $O/t9_pytest.log
