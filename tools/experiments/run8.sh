cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
python -m pytest tests/test_train_step.py -m gpu -x -q -k "graphed or own_forward" -s 2>&1 | grep -a "passed\|failed\|Error\|error\|grad_x\|magnitude" | tail -60 > gpurun_out/r6/t_gts.log; tail -45 gpurun_out/r6/t_gts.log
python bench.py > gpurun_out/r6/bench1.json 2> gpurun_out/r6/bench1.err; tail -3 gpurun_out/r6/bench1.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/r6/bench1.json'))
print(json.dumps(r['e2e']['frozen'], indent=1)[:3000])
PY
timeout 1500 python tests/proxy_ap.py --images 256 --out gpurun_out/r6/proxy_ap.json > gpurun_out/r6/proxy_ap.log 2>&1; tail -3 gpurun_out/r6/proxy_ap.log | cut -c1-3000
