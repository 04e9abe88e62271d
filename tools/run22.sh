#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 2400 python3 -m pytest tests/test_gpu_deferred.py tests/test_gpu_knobs.py tests/test_gpu_faults.py tests/test_gpu_configs.py tests/test_gpu_parity.py -q -x 2>&1 | tail -4
ROUNDS=5 bash tools/ab.sh "" "RN_PENDING_LT=0" "RN_PENDING_LT=1" 2>&1 | cut -c1-330
