"""MFMA utilisation per kernel from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (kernel-trace only).
   util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), the busy cycles as the sum over
the 1024 SIMDs (256 CUs x 4): the quotient is the fraction of SIMD-cycles in which an MFMA was executing while the kernel
ran — the dense MFMA peak corresponds to 1.0.   python tools/pmc_mfma.py <dir> [min share of busy cycles to list]"""
import csv, glob, json, sys
from collections import defaultdict
d = sys.argv[1]
f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
busy, act, n = defaultdict(float), defaultdict(float), defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if k.startswith("set_u32_kernel"):      # bench.py's empty-kernel calibration launches, not part of the step
        continue
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        act[k] += float(r["Counter_Value"])
tot_b, tot_a = sum(busy.values()), sum(act.values())
rows = []
for k in sorted(busy, key=lambda x: -act[x]):
    if act[k] <= 0:
        continue
    rows.append({"kernel": k[:90], "launches": len(n[k]), "share_of_gpu_active_cycles": round(act[k] / tot_a, 4),
                 "mfma_util": round(busy[k] / (act[k] / 8.0 * 1024.0), 4)})
out = {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)", "all_kernels_mfma_util": round(tot_b / (tot_a / 8.0 * 1024.0), 4),
       "note": "kernels overlap on two streams, so per-kernel active cycles add up to more than the step's; the all-kernel figure is "
               "busy cycles over the SUM of per-kernel active cycles, a lower bound of the step's utilisation",
       "kernels": [r for r in rows if r["share_of_gpu_active_cycles"] >= float(sys.argv[2] if len(sys.argv) > 2 else 0.005)]}
print(json.dumps(out, indent=1))
