"""Busy time vs span of the last pass in a rocprofv3 --kernel-trace CSV (kernels may overlap on several streams):
tools/trace_busy.py <kernel_trace.csv> <first kernel name fragment>.  Prints span, union of busy intervals, idle
gaps and the largest gaps with the kernels around them."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2]
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
last = rows[starts[-1]:]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0][:36]) for r in last)
span = (max(e for _, e, _ in iv) - iv[0][0]) / 1e3
busy, gaps, cur_s, cur_e, prev_name = 0.0, [], iv[0][0], iv[0][1], iv[0][2]
for s, e, n in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(((s - cur_e) / 1e3, prev_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev_name = n
busy += cur_e - cur_s
print("kernels %d  span %.1f us  busy %.1f us  idle %.1f us in %d gaps" % (len(iv), span, busy / 1e3, span - busy / 1e3, len(gaps)))
for g in sorted(gaps, reverse=True)[:12]:
    print("  gap %6.1f us  after %-36s before %s" % g)
print("sum of kernel durations %.1f us" % (sum(e - s for s, e, _ in iv) / 1e3))
