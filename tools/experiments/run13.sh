cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r6/gputest_full.log; cat gpurun_out/r6/gputest_full.log
