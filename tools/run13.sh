#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 1800 python3 -m pytest tests -m gpu -q > $O/t13_full.log 2>&1; echo "full rc=$?" >> $O/t13_full.log; tail -4 $O/t13_full.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
