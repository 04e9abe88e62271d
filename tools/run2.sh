#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_faults.py > $O/t2_pytest.log 2>&1; echo "pytest rc=$?" >> $O/t2_pytest.log
timeout 600 python3 -m pytest tests/test_gpu_faults.py -q > $O/t2_faults.log 2>&1; echo "faults rc=$?" >> $O/t2_faults.log
rocprofv3 -L > $O/counters.txt 2>&1
S="3100 6144 1024 0 0 1"
for impl in ours lib; do
 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq_$impl -o x --output-format csv -- python3 tools/gemm_one.py $S $impl > /dev/null 2>&1
 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d $O/pmc_tcc_$impl -o x --output-format csv -- python3 tools/gemm_one.py $S $impl > /dev/null 2>&1
 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$impl -o x --output-format csv -- python3 tools/gemm_one.py $S $impl > /dev/null 2>&1
done
python3 - <<'PY' > gpurun_out/r05/pmc_gemm_summary.txt 2>&1
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r05/pmc_*_*")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k in acc:
        n = len(cnt[k])
        if n < 5: continue
        print(d.split("/")[-1], "|", k, "| launches", n, "|", "  ".join("%s=%.4g" % (c, v / n) for c, v in sorted(acc[k].items())))
PY
rm -rf $O/pmc_sq_* $O/pmc_tcc_* $O/pmc_fetch_*
tail -3 $O/t2_pytest.log; tail -3 $O/t2_faults.log; cat $O/pmc_gemm_summary.txt
