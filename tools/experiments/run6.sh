cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for side in 0 1; do
  for rep in 1 2; do
    echo "== capture on side stream: $side (run $rep)"
    CDN_GTS_CAPTURE_ON_SIDE=$side timeout 600 python tools/experiments/gts_probe2.py whole 2>&1 | grep -a "GTS2\|round 0\|Error\|error\|AccumulateGrad" | cut -c1-260
  done
done > gpurun_out/r6/gts_probe6.log 2>&1
cat gpurun_out/r6/gts_probe6.log
python -m pytest tests/test_train_step.py -m gpu -x -q -k "own_forward_decisions or 2_and_3_bit" 2>&1 | tail -40 > gpurun_out/r6/t_twin.log; cat gpurun_out/r6/t_twin.log
