cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
(timeout 600 python tools/experiments/gts_probe2.py whole) 2>&1 | grep -a "GTS2\|round 0\|Error\|error\|loss\|tensors" > gpurun_out/r6/gts_probe4.log
cat gpurun_out/r6/gts_probe4.log
