#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
ROUNDS=6 bash tools/ab.sh "--rec local --batch 32 --frames 40 --feat 2048" "RN_BLASLT=0" "RN_BLASLT=1" 2>&1 | cut -c1-400
ROUNDS=5 bash tools/ab.sh "--rec local" "RN_BLASLT=0" "RN_BLASLT=1" 2>&1 | cut -c1-160
