#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
RN_LIB_PROBE=1 python3 tools/loc_chain_probe.py 100 28 1536 global 2>&1 | tail -17
ROUNDS=5 bash tools/ab.sh "" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT=" 2>&1 | cut -c1-420
ROUNDS=3 bash tools/ab.sh "--rec local" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT=" 2>&1 | cut -c1-160
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_softmax.py tests/test_gpu_acquire_inv.py -q -x > $O/t14_pytest.log 2>&1; echo "pytest rc=$?" >> $O/t14_pytest.log
tail -3 $O/t14_pytest.log
