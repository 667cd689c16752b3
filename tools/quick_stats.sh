#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the default bench (no CPU / whole-network legs); prints the top kernels.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/qs && mkdir -p gpurun_out/qs
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/qs -- python3 bench.py --no-cpu-baseline --no-e2e --no-config-legs "$@" > gpurun_out/qs.log 2>&1
f=$(find gpurun_out/qs -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void (anonymous namespace)::", "")
    print("%-60s calls %5s avg %8.1f us  min %8.1f  max %8.1f" % (name[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
find gpurun_out/qs -name "*kernel_trace.csv" -delete
