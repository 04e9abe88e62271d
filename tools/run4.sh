#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
K='test_deferred_update_equals_the_immediate_one_once_flushed and None-recurrent-full-chains-global-bf16'
echo "== A: as is" > $O/t4.log
timeout 300 python3 -X faulthandler -m pytest tests/test_gpu_deferred.py -q -x -k "$K" >> $O/t4.log 2>&1; echo "A rc=$?" >> $O/t4.log
echo "== B: no images_stale" >> $O/t4.log
RN_T_SKIP_IMAGES=1 timeout 300 python3 -X faulthandler -m pytest tests/test_gpu_deferred.py -q -x -k "$K" >> $O/t4.log 2>&1; echo "B rc=$?" >> $O/t4.log
echo "== C: gdb backtrace" >> $O/t4.log
timeout 600 /opt/rocm/bin/rocgdb -batch -ex "handle SIGSEGV stop nopass" -ex run -ex bt -ex "info threads" --args python3 -m pytest tests/test_gpu_deferred.py -q -x -k "$K" >> $O/t4.log 2>&1
grep -E "rc=|==|#[0-9]+ " $O/t4.log | head -60
timeout 300 python3 tools/gemm_cold_probe.py > $O/gemm_cold2.txt 2>&1; cat $O/gemm_cold2.txt
timeout 600 python3 -m pytest tests/test_gpu_gemm.py -q > $O/t4_gemm.log 2>&1; tail -5 $O/t4_gemm.log
