#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the QAT step (tools/train_step_bench.py, eager so that kernels carry names).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ts && mkdir -p gpurun_out/ts
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ts -- python3 tools/train_step_bench.py --steps 20 "$@" > gpurun_out/ts.log 2>&1
f=$(find gpurun_out/ts -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/train_step_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
nat = sum(float(r["TotalDurationNs"]) for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"])
print("total %.1f ms, at::native + copies %.1f %%" % (tot / 1e6, 100 * nat / tot))
for r in rows[:28]:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"]).split("(")[0]
    print("%-70s calls %5s avg %8.1f us  %5.1f %%" % (name[:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -1 gpurun_out/ts.log
find gpurun_out/ts -name "*kernel_trace.csv" -delete
