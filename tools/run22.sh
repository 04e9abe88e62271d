#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
ROUNDS=6 bash tools/ab.sh "" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT=" 2>&1 | cut -c1-150
ROUNDS=3 bash tools/ab.sh "--rec local" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT=" 2>&1 | cut -c1-150
ROUNDS=3 bash tools/ab.sh "--rec none" "RN_LIB_VARIANT=base" "RN_LIB_VARIANT=" 2>&1 | cut -c1-150
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
