"""Grid fill of every kernel of one steady-state iteration, from a rocprofv3 --kernel-trace CSV (argv[1]; the iteration is
the span between the last two launches of `marker`, argv[2], default stem_kernel): workgroups, workgroups per CU (256
CUs), threads per workgroup, static LDS, average duration.  What found the half-empty gather grids of cfg2 and layer 4's
one-wave-per-SIMD pointwise launches."""
import collections
import csv
import sys

f = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "stem_kernel"
rows = list(csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
seen = collections.OrderedDict()
for r in rows[a:b]:
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    g = [int(r["Grid_Size_" + d]) for d in "XYZ"]
    w = [int(r["Workgroup_Size_" + d]) for d in "XYZ"]
    wgs = (g[0] // w[0]) * (g[1] // w[1]) * (g[2] // w[2])
    threads = w[0] * w[1] * w[2]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = (n, wgs, threads)
    if key not in seen:
        seen[key] = [0, 0.0, int(r.get("LDS_Block_Size", 0) or 0)]
    seen[key][0] += 1
    seen[key][1] += dur
for (n, wgs, threads), (cnt, dur, lds) in seen.items():
    print("%-44s x%2d  workgroups %6d = %6.2f per CU  x %4d threads  static LDS %6d  avg %6.1f us"
          % (n, cnt, wgs, wgs / 256.0, threads, lds, dur / cnt))
