"""Per-kernel totals of the steady-state part of a rocprofv3 kernel trace: everything after the last kernel whose name
matches one of the given 'setup' substrings (MIOpen / rocBLAS kernels of a calibration pass), per iteration.
    python tools/steady_stats.py <fz_kernel_trace.csv> <iterations | marker:<kernel substring>> [setup substrings ...]
(marker:<s>: the iteration count is the number of steady-state dispatches whose name contains <s> -- one per iteration)"""
import csv, sys, collections
path, iters = sys.argv[1], sys.argv[2]
setup = sys.argv[3:] or ["naive_conv", "miopen", "Cijk_", "Im2d2Col", "kernel_grouped_conv"]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
last = max([i for i, r in enumerate(rows) if any(s in r["Kernel_Name"] for s in setup)] or [-1])
rows = rows[last + 1:]
iters = sum(1 for r in rows if iters[7:] in r["Kernel_Name"]) if iters.startswith("marker:") else int(iters)
tot, cnt = collections.Counter(), collections.Counter()
for r in rows:
    n = r["Kernel_Name"]
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0][:60]
    tot[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[n] += 1
total = sum(tot.values())
print("steady state: %d kernels, %.3f ms of kernel time per iteration (%d iterations)" % (len(rows), total / 1e6 / iters, iters))
for n, t in tot.most_common(40):
    print("  %-60s %6.1f launches/iter  avg %8.1f us   %7.3f ms/iter  %5.1f%%" % (n, cnt[n] / iters, t / cnt[n] / 1e3, t / 1e6 / iters, 100.0 * t / total))
