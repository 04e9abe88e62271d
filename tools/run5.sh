#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
./tools/micro/gemm_probe > $O/gemm_probe.txt 2>&1; cat $O/gemm_probe.txt
echo "== deferred file, no images_stale" > $O/t5.log
RN_T_SKIP_IMAGES=1 timeout 900 python3 -X faulthandler -m pytest tests/test_gpu_deferred.py -q -x >> $O/t5.log 2>&1; echo "skip-images rc=$?" >> $O/t5.log
echo "== deferred file, with images_stale under gdb" >> $O/t5.log
timeout 900 /opt/rocm/bin/rocgdb -batch -ex "handle SIGSEGV stop nopass" -ex run -ex bt --args python3 -m pytest tests/test_gpu_deferred.py -q -x >> $O/t5.log 2>&1; echo "gdb rc=$?" >> $O/t5.log
grep -E "rc=|==|^#[0-9]+ |passed|failed|SIGSEGV" $O/t5.log | head -60
