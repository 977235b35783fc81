"""Timeline of one bench step out of a `rocprofv3 --kernel-trace` csv: every dispatch between two
consecutive drifts (k_drift), times in microseconds from the step's first kernel.
  python tests/rocpd_timeline.py <kernel_trace.csv> [step_from_end] [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_drift')]
i0, i1 = idx[-back - 1], idx[-back]
t0 = int(rows[i0]['Start_Timestamp'])
print("step length %.1f us, %d dispatches" % ((int(rows[i1]['Start_Timestamp']) - t0) / 1e3, i1 - i0))
last_end = {}
for r in rows[i0:i1]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    e = (int(r['End_Timestamp']) - t0) / 1e3
    nm = r['Kernel_Name'].split('(')[0]
    nm = nm.replace('void ', '').replace('rocprim::ROCPRIM_400200_NS::detail::', 'rp::')[:70]
    q = r.get('Queue_Id')
    gap = s - last_end.get(q, s)
    last_end[q] = e
    if e - s >= min_us:
        print("%8.1f %8.1f %7.1f gap %6.1f q%s %s" % (s, e, e - s, gap, q, nm))
