#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
ROUNDS=5 bash tools/ab.sh "" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT=" 2>&1 | cut -c1-330 | tee $O/ab_sumsq.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_loop.py tests/test_gpu_fullsize.py -q -x > $O/t8_pytest.log 2>&1; echo "pytest rc=$?" >> $O/t8_pytest.log
tail -4 $O/t8_pytest.log
