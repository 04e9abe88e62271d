#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 900 python3 -m pytest tests -m gpu -q -x --deselect tests/test_gpu_faults.py > $O/t1_pytest.log 2>&1; echo "pytest rc=$?" >> $O/t1_pytest.log
timeout 600 python3 -m pytest tests/test_gpu_faults.py -q > $O/t1_faults.log 2>&1; echo "faults rc=$?" >> $O/t1_faults.log
timeout 300 python3 bench.py > $O/bench_c2_run1.json 2> $O/bench_c2_run1.err
timeout 300 python3 tools/gemm_cold_probe.py > $O/gemm_cold.txt 2>&1
tail -5 $O/t1_pytest.log; tail -5 $O/t1_faults.log; cat $O/gemm_cold.txt
