"""Print the kernels of the LAST pass of a rocprofv3 --kernel-trace CSV in launch order with durations
(tools/trace_order.py <kernel_trace.csv> <kernels per pass or 0 = detect by first kernel name>)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = rows[0]["Kernel_Name"] if len(sys.argv) < 3 else sys.argv[2]
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# last complete pass
last = rows[starts[-1]:]
t_end_prev = None
tot = 0.0
for r in last:
    n = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = 0.0 if t_end_prev is None else (int(r["Start_Timestamp"]) - t_end_prev) / 1e3
    t_end_prev = int(r["End_Timestamp"])
    tot += d
    print("%-44s grid %8s wg %5s  %8.1f us  gap %6.1f" % (n[:44], r.get("Grid_Size_X", r.get("Grid_Size", "?")),
                                                          r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), d, gap))
print("sum %.1f us, span %.1f us" % (tot, (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3))
