cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heads" 2>&1 | tail -8
python -m pytest tests/test_harness.py tests/test_gpu_backbone.py -m gpu -x -q 2>&1 | tail -3
python tools/heads_bench.py 2>/dev/null | cut -c1-300
python tools/e2e_native_bench.py --graph 2>/dev/null | cut -c1-200
python tools/e2e_native_bench.py --graph 2>/dev/null | cut -c1-200
