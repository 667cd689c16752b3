# On the GPU box: kernel stats of tools/decode_bench.py for the shipped library and variant builds (A/B on one box):
#   bash tools/dec_prof.sh "" _variant ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for v in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/dec$v -- python3 tools/with_lib.py codenet_amd/lib/libcodenet_dcn$v.so tools/decode_bench.py > gpurun_out/r04/dec$v.log 2>&1
  f=$(find gpurun_out/r04/dec$v -name "*kernel_stats.csv" | head -1)
  echo "variant '$v' $(tail -1 gpurun_out/r04/dec$v.log)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "decode_" in r["Name"] or "sigmoid_store" in r["Name"]:
        print("   %-28s calls %s avg %.1f us" % (r["Name"].split("::")[-1][:28], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  find gpurun_out/r04/dec$v -name "*.csv" -delete
done
