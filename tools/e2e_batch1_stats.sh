cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/b1 && mkdir -p gpurun_out/b1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b1 -- python3 tools/e2e_native_bench.py --batch 1 --steps 50 > gpurun_out/b1.log 2>&1
f=$(find gpurun_out/b1 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    if int(r["Calls"]) % 50 == 0 or int(r["Calls"]) >= 50:
        tot += float(r["TotalDurationNs"])
print("total us per step (kernels with >= 50 calls):", tot / 50 / 1e3)
for r in rows[:28]:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"]).split("(")[0]
    print("%-46s calls %5s avg %7.1f us  total/step %7.1f" % (name[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 50 / 1e3))
PY
find gpurun_out/b1 -name "*kernel_trace.csv" -delete
