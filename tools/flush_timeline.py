import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "advance_step" in r[0]]
# the timed loop's last step = the step before the first big gap after a run of replays; print everything from the last advance_step of the first dense run
# find runs: consecutive advance_steps less than 3 ms apart
runs = []; cur = [idx[0]]
for a, b in zip(idx, idx[1:]):
    if rows[b][1] - rows[a][1] < 3e6: cur.append(b)
    else: runs.append(cur); cur = [b]
runs.append(cur)
run = max(runs[:2], key=len) if len(runs) > 1 else runs[0]
run = runs[0] if len(runs[0]) >= 4 else run
last = run[-1]
t0 = rows[last][1]
for n, s, e, q in rows[last:last + 90]:
    if (s - t0) / 1e3 > 2600: break
    nm = re.sub(r"\s+", " ", n).split("(")[0].replace("void ", "")
    if (s - t0) / 1e3 > 1350: print("%8.1f +%7.1f us q%-2d %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, nm[:70]))
