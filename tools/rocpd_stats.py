#!/usr/bin/env python3
"""Per-kernel statistics (the `--stats` table) from a rocprofv3 rocpd database (`*_results.db`), as CSV:
   python tools/rocpd_stats.py gpurun_out/prof_c3/c3_results.db [steps] > profiles/r02_kernel_stats_c3.csv
`steps` (optional): number of train steps the run executed, adds a per-step column."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = db.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                      "avg(1.0*(end-start)*(end-start)) from kernels group by %s order by 3 desc" % (name, name)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    hdr = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDevNs"] + (["UsPerStep", "CallsPerStep"] if steps else [])
    print(",".join(hdr))
    for n, c, s, a, mn, mx, a2 in rows:
        n = re.sub(r"\s+", " ", n)
        if len(n) > 160:
            n = n[:157] + "..."
        out = ['"%s"' % n.replace('"', "'"), str(c), str(int(s)), "%.1f" % a, "%.2f" % (100.0 * s / tot), str(int(mn)), str(int(mx)), "%.1f" % (max(a2 - a * a, 0.0) ** 0.5)]
        if steps:
            out += ["%.1f" % (s / steps / 1e3), "%.2f" % (c / steps)]
        print(",".join(out))


if __name__ == "__main__":
    main()
