cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
(timeout 600 python tools/experiments/gts_probe2.py whole; timeout 600 python tools/experiments/gts_probe2.py whole; timeout 300 python tools/experiments/gts_probe2.py stack) 2>&1 | grep -a "GTS2\|round 0\|Error\|error" > gpurun_out/r6/gts_probe3.log
cat gpurun_out/r6/gts_probe3.log
