import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the last full step: find last occurrence of advance_step_kernel
idx = [i for i, r in enumerate(rows) if "advance_step" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"]); t1 = int(rows[b]["Start_Timestamp"])
print("step wall (start to next start): %.1f us, kernels: %d" % ((t1 - t0) / 1e3, len(step)))
dur = collections.defaultdict(float); gap = collections.defaultdict(float); cnt = collections.Counter()
prev_end = None
for r in step:
    n = r["Kernel_Name"]
    n = n.split("(")[0][:58]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n] += e - s; cnt[n] += 1
    if prev_end is not None: gap[n] += max(0, s - prev_end)
    prev_end = max(prev_end or 0, e)
tot_d = sum(dur.values()); tot_g = sum(gap.values())
print("sum of durations %.1f us, sum of gaps before kernels %.1f us" % (tot_d / 1e3, tot_g / 1e3))
for n in sorted(dur, key=lambda n: -(dur[n] + gap[n]))[:22]:
    print("%-58s n=%3d dur=%7.1f us (avg %5.2f) gap_before=%6.1f us (avg %4.2f)" % (n, cnt[n], dur[n] / 1e3, dur[n] / cnt[n] / 1e3, gap[n] / 1e3, gap[n] / cnt[n] / 1e3))
