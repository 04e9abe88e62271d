#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do timeout 600 python3 -m pytest tests/test_gpu_deferred.py -q -x -k "row_groups and global and f32" 2>&1 | grep -E "passed|failed|assert|Error" | head -8; done
echo "== base"
for i in 1 2; do RN_LIB_VARIANT=base timeout 600 python3 -m pytest tests/test_gpu_deferred.py -q -x -k "row_groups and global and f32" 2>&1 | grep -E "passed|failed|assert|Error" | head -8; done
