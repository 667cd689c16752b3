cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_real_shapes.py -m gpu -x -q -k "cfg2 or chained or fp32" 2>&1 | tail -8
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
python bench.py --config cfg2 > gpurun_out/r6/bench_cfg2.json 2> gpurun_out/r6/bench_cfg2.err; tail -2 gpurun_out/r6/bench_cfg2.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/r6/bench_cfg2.json'))
print(r['ms_per_step'], r['value'], r.get('kernel_ms_per_launch'))
PY
python bench.py --fp32 > gpurun_out/r6/bench_fp32.json 2> gpurun_out/r6/bench_fp32.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/r6/bench_fp32.json'))
print(r['ms_per_step'], r['value'], r.get('kernel_ms_per_launch'))
PY
