cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
python -m pytest tests/test_train_step.py -m gpu -x -q -k "graphed" 2>&1 | tail -5
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generic" 2>&1 | tail -5
python tools/generic_bench.py > gpurun_out/r6/generic_bench.json 2> gpurun_out/r6/generic_bench.err; tail -2 gpurun_out/r6/generic_bench.err; cat gpurun_out/r6/generic_bench.json | cut -c1-1500
python tools/serving_margin_sweep.py > gpurun_out/r6/serving_margin_sweep.json 2> gpurun_out/r6/serving_margin_sweep.err; grep margin gpurun_out/r6/serving_margin_sweep.err
