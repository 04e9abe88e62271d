#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
ROUNDS=5 bash tools/ab.sh "" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT=" 2>&1 | cut -c1-330
ROUNDS=3 bash tools/ab.sh "--rec local" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT=" 2>&1 | cut -c1-200
timeout 900 python3 -m pytest tests/test_gpu_deferred.py tests/test_gpu_knobs.py tests/test_gpu_configs.py -q -x 2>&1 | tail -3
