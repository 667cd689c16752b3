cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
(timeout 600 python tools/experiments/graph_reduce_repro.py 4000; timeout 600 python tools/experiments/graph_reduce_repro.py 4000) 2>&1 | grep -a "GRAPH_REDUCE\|Error" > gpurun_out/r6/graph_reduce.log
cat gpurun_out/r6/graph_reduce.log
python -m pytest tests/test_train_step.py -m gpu -x -q -k "own_forward_decisions or 2_and_3_bit" 2>&1 | tail -40 > gpurun_out/r6/t_twin.log; cat gpurun_out/r6/t_twin.log
