#!/usr/bin/env python3
"""Timeline of ONE replayed train step from a rocprofv3 rocpd database: every kernel of the last complete step in start
order with its offset from the step's start, duration and queue — shows what sits on the critical path between the chain
kernels and what overlaps.   python tools/step_timeline.py gpurun_out/prof_x/x_results.db [min_us]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "advance_step" in r[0]]
a, b = idx[-2], idx[-1]
t0 = rows[a][1]
print("step: %.1f us from advance_step to the next advance_step, %d kernels" % ((rows[b][1] - t0) / 1e3, b - a))
# how much of the step has NO kernel running on any queue (launch / dependency latency), and the sum of kernel run times
idle_all, busy_sum, fr = 0.0, 0.0, t0
for n, s, e, q in rows[a:b]:
    if s > fr:
        idle_all += (s - fr) / 1e3
    fr = max(fr, e)
    busy_sum += (e - s) / 1e3
idle_all += max(0.0, (rows[b][1] - fr) / 1e3)
print("no kernel running on any queue: %.1f us of the step (%.1f %%); sum of kernel run times %.1f us" % (
    idle_all, 100.0 * idle_all / ((rows[b][1] - t0) / 1e3), busy_sum))
frontier = t0
for n, s, e, q in rows[a:b]:
    nm = re.sub(r"\s+", " ", n).split("(")[0]
    nm = nm.replace("void ", "")
    gap = (s - frontier) / 1e3
    if (e - s) / 1e3 >= min_us:
        print("%8.1f  +%7.1f us  q%-2d %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, nm[:70], ("   [idle %.1f us before]" % gap) if gap > 3 else ""))
    frontier = max(frontier, e)
