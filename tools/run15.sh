#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
ROUNDS=4 bash tools/ab.sh "--rec local" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT=" 2>&1 | cut -c1-420
ROUNDS=2 bash tools/ab.sh "--rec local --batch 32 --frames 40 --feat 2048" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT=" 2>&1 | cut -c1-420
timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_acquire_inv.py tests/test_gpu_deferred.py -q -x > $O/t15_pytest.log 2>&1; echo "pytest rc=$?" >> $O/t15_pytest.log
tail -3 $O/t15_pytest.log
