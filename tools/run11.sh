#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_deferred.py -q -x > $O/t11_plain.log 2>&1; echo "plain rc=$?" >> $O/t11_plain.log
tail -3 $O/t11_plain.log
timeout 1200 /opt/rocm/bin/rocgdb -batch -ex "handle SIGSEGV stop nopass" -ex run -ex "bt 40" -ex "info sharedlibrary" --args python3 -m pytest tests/test_gpu_deferred.py -q -x > $O/t11_gdb.log 2>&1; echo "gdb rc=$?" >> $O/t11_gdb.log
grep -n "SIGSEGV\|^#[0-9]" $O/t11_gdb.log | head -50
tail -3 $O/t11_gdb.log
