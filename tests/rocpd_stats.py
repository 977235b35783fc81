"""Summarise a rocprofv3 rocpd database (kernel trace) as the --stats CSV would: per-kernel calls,
total/average/min/max duration.  Usage: python tests/rocpd_stats.py results.db [out.csv] [--timeline N]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = db.execute("select %s, start, end from kernels order by start" % name).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
    tot = sum(a[1] for a in agg.values())
    out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append((n, a[0], a[1], "%.1f" % (a[1] / a[0]), "%.4f" % (100.0 * a[1] / tot), a[2], a[3]))
    if len(sys.argv) > 2 and not sys.argv[2].startswith("--"):
        with open(sys.argv[2], "w", newline="") as f:
            csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(out)
    else:
        for r in out[:60]:
            print("%-70s %6s %12s %12s %8s" % (r[0][:70], r[1], r[2], r[3], r[4]))
    if "--timeline" in sys.argv:
        k = int(sys.argv[sys.argv.index("--timeline") + 1])
        # the last k kernels before the last gravity walk: one tree build
        idx = [i for i, r in enumerate(rows) if r[0].startswith("void k_grav_walk<0")]
        last = idx[-1]
        t0 = rows[last - k][1]
        for n, s, e in rows[last - k:last + 1]:
            print("%9.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:90]))


main()
