#!/bin/bash
# On the GPU box: per-kernel averages (rocprofv3 kernel stats of the default bench) for the committed library and
# for each variant library given:  bash tools/ab_stats.sh <pattern> codenet_amd/lib/libcodenet_dcn_<tag>.so ...
PAT="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
one() {
  rm -rf gpurun_out/abs && mkdir -p gpurun_out/abs
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abs -- python3 tools/with_lib.py "$1" bench.py --no-cpu-baseline --no-e2e > gpurun_out/abs.log 2>&1
  f=$(find gpurun_out/abs -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$PAT" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"]).split("(")[0]
        print("   %-44s calls %4s avg %7.1f us  min %7.1f  max %7.1f" % (name[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  find gpurun_out/abs -name "*kernel_trace.csv" -delete
}
echo base; one "$GRAFT_REPO_ROOT/codenet_amd/lib/libcodenet_dcn.so"
for v in "$@"; do echo "$v"; one "$GRAFT_REPO_ROOT/$v"; done
