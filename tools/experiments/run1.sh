cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
python -m pytest tests/test_train_step.py -m gpu -x -q > gpurun_out/r6/t_train.log 2>&1; echo "train tests rc=$?" 
for m in alone alone repro poison b_stack b_torch b_fwd b_fwd_grad b_noopt b_after; do
  timeout 300 python tools/experiments/gts_probe.py $m 2>&1 | grep -a "GTS\|Error\|error" | tail -3
done > gpurun_out/r6/gts_probe.log 2>&1
python bench.py > gpurun_out/r6/bench0.json 2> gpurun_out/r6/bench0.err
tail -3 gpurun_out/r6/t_train.log; cat gpurun_out/r6/gts_probe.log
