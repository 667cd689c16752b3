cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for m in alone b_torch b_fwd b_noopt b_adam_tiny b_sgd_tiny sleep; do
  timeout 300 python tools/experiments/gts_probe.py $m --steps 3 2>&1 | grep -a "GTS\|Error\|error" | tail -3
done > gpurun_out/r6/gts_probe2.log 2>&1
cat gpurun_out/r6/gts_probe2.log
