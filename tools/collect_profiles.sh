#!/bin/bash
# Runs on the GPU box (via gpurun): bench variants, rocprofv3 kernel stats and the PMC passes
# (separate passes, no trace domains combined with --pmc), all into gpurun_out/$1/.
# Afterwards: python tools/summarize_profiles.py $1   (in the build container) -> profiles/$1/
set -u
R=${1:-r01}
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err
python3 bench.py --frozen --no-cpu-baseline > $OUT/bench_frozen.json 2>/dev/null
python3 bench.py --fp32 --no-cpu-baseline > $OUT/bench_fp32.json 2>/dev/null
python3 bench.py --w2 --no-cpu-baseline > $OUT/bench_w2.json 2>/dev/null
python3 bench.py --res 256 --batch 32 --fp32 --no-cpu-baseline > $OUT/bench_cfg2.json 2>/dev/null
python3 bench.py --path modules --no-cpu-baseline > $OUT/bench_modules.json 2>/dev/null
python3 tools/e2e_bench.py > $OUT/e2e_w4a8.json 2>/dev/null
python3 tools/train_step_bench.py > $OUT/train_step_w4a8.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1
# rows f1-f4 (heads, decode, backbone, whole network): bench lines, kernel stats and the backbone's launch-order trace
python3 tools/heads_bench.py > $OUT/heads_bench.json 2>/dev/null
python3 tools/decode_bench.py > $OUT/decode_bench.json 2>/dev/null
python3 tools/backbone_bench.py > $OUT/backbone_bench.json 2>/dev/null
python3 tools/e2e_native_bench.py --graph > $OUT/e2e_native.json 2>/dev/null
python3 tools/e2e_bench.py --fp32 > $OUT/e2e_fp32.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e_stats -- python3 tools/e2e_native_bench.py > $OUT/e2e_stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/bb_trace -- python3 tools/backbone_bench.py > $OUT/bb_trace.log 2>&1
f=$(find $OUT/bb_trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_order.py $f stem_kernel > $OUT/backbone_kernel_order.txt
f=$(find $OUT/e2e_stats -name "*kernel_stats.csv" | head -1)
cp $f $OUT/e2e_native_kernel_stats.csv
rm -rf $OUT/bb_trace $OUT/e2e_stats/*/*kernel_trace.csv
ls $OUT
