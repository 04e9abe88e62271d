#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
ROUNDS=3 bash tools/ab.sh "" "RN_DHID_FUSED=0 RN_JOIN_EARLY=0" "RN_DHID_FUSED=1 RN_JOIN_EARLY=0" "RN_DHID_FUSED=0 RN_JOIN_EARLY=1" "RN_DHID_FUSED=1 RN_JOIN_EARLY=1" 2>&1 | cut -c1-600 | tee $O/ab_dhid.txt
ROUNDS=2 bash tools/ab.sh "--rec local" "RN_JOIN_EARLY=0" "RN_JOIN_EARLY=1" 2>&1 | cut -c1-600 | tee -a $O/ab_dhid.txt
timeout 1500 python3 -m pytest tests -m gpu -q -x --deselect tests/test_gpu_faults.py > $O/t6_pytest.log 2>&1; echo "pytest rc=$?" >> $O/t6_pytest.log
tail -6 $O/t6_pytest.log
