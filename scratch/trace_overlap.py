import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "advance_step" in r["Kernel_Name"]]
step = rows[idx[-2]:idx[-1]]
ov = 0; tot = 0; n_ov = 0
queues = set()
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); tot += e - s
    queues.add(r.get("Queue_Id"))
    for q in step[i+1:i+6]:
        s2, e2 = int(q["Start_Timestamp"]), int(q["End_Timestamp"])
        o = min(e, e2) - max(s, s2)
        if o > 0: ov += o; n_ov += 1
print("kernels", len(step), "queues", queues, "sum dur %.1f us" % (tot/1e3), "pairwise overlap %.1f us in %d pairs" % (ov/1e3, n_ov))
print("wall %.1f us" % ((int(rows[idx[-1]]["Start_Timestamp"]) - int(step[0]["Start_Timestamp"]))/1e3))
