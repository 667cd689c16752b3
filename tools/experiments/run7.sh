cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for two in 1 1 0; do
    echo "== two-level loss reduction: $two"
    GTS_TWO_LEVEL=$two timeout 600 python tools/experiments/gts_probe2.py whole 2>&1 | grep -a "GTS2\|round 0\|Error\|error" | cut -c1-260
done > gpurun_out/r6/gts_probe7.log 2>&1
cat gpurun_out/r6/gts_probe7.log
python tools/with_lib.py codenet_amd/lib/libcodenet_dcn_stamps.so tools/pw_bench.py 65536 232 116 232 > gpurun_out/r6/pwd3_stamps.log 2>&1
python tools/with_lib.py codenet_amd/lib/libcodenet_dcn_stamps.so tools/pw_bench.py 262144 116 58 116 >> gpurun_out/r6/pwd3_stamps.log 2>&1
cat gpurun_out/r6/pwd3_stamps.log
