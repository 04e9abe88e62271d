#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for m in 0 1; do echo "== RN_DEC_PARTIAL=$m"; RN_DEC_PARTIAL=$m RN_LIB_PROBE=1 python3 tools/loc_chain_probe.py 100 28 1536 global 2>&1 | grep -A 18 "dec fwd"; done
