cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
python -m pytest tests/test_train_step.py -m gpu -x -q 2>&1 | tail -6
for i in 1 2; do
python tools/train_step_bench.py --graph --no-fused-update 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('launches:', r.get('ms_per_step'), r.get('ms'))"
python tools/train_step_bench.py --graph 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('fused   :', r.get('ms_per_step'), r.get('ms'))"
done
