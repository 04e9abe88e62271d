#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
ROUNDS=5 bash tools/ab.sh "" "RN_BLASLT=0" "RN_BLASLT=1" 2>&1 | cut -c1-330
ROUNDS=3 bash tools/ab.sh "--rec local" "RN_BLASLT=0" "RN_BLASLT=1" 2>&1 | cut -c1-330
ROUNDS=3 bash tools/ab.sh "--rec local --batch 64 --frames 28 --feat 3584" "RN_BLASLT=0" "RN_BLASLT=1" 2>&1 | cut -c1-330
ROUNDS=2 bash tools/ab.sh "--rec local --batch 32 --frames 40 --feat 2048" "RN_BLASLT=0" "RN_BLASLT=1" 2>&1 | cut -c1-130
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x 2>&1 | tail -3
