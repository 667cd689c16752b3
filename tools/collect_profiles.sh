#!/bin/bash
# Runs on the GPU box (via gpurun): bench variants, rocprofv3 kernel stats and the PMC passes
# (separate passes, no trace domains combined with --pmc), all into gpurun_out/<round>/.
# Afterwards: python tools/summarize_profiles.py <round>   (in the build container) -> profiles/<round>/
set -u
R=${1:-r06}
OUT="gpurun_out/$R"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py"
$B > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"
$B --frozen --no-cpu-baseline > "$OUT/bench_frozen.json" 2>/dev/null
$B --fp32 --no-cpu-baseline > "$OUT/bench_fp32.json" 2>/dev/null
$B --config cfg4 --no-cpu-baseline > "$OUT/bench_cfg4.json" 2>/dev/null
$B --config cfg2 --no-cpu-baseline > "$OUT/bench_cfg2.json" 2>/dev/null
$B --path modules --no-cpu-baseline > "$OUT/bench_modules.json" 2>/dev/null
python3 tools/train_step_bench.py > "$OUT/train_step_w4a8_eager.json" 2>/dev/null
python3 tools/train_step_bench.py --graph > "$OUT/train_step_w4a8.json" 2>/dev/null
Q="--no-cpu-baseline --no-e2e --no-config-legs"
# kernel stats: running (reference-faithful) schedule with its frozen leg, and the frozen schedule as the step
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $B $Q > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_frozen" -- $B --frozen $Q > "$OUT/stats_frozen.log" 2>&1
P="--steps 3 --warmup 2 --no-graph $Q"
MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
for mode in running frozen; do
  F=""; [ $mode = frozen ] && F="--frozen"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$mode" -- $B $F $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$mode" -- $B $F $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_sq_$mode" -- $B $F $P > /dev/null 2>&1
  # matrix-core counters: busy cycles of the MFMA pipes against the CUs' busy cycles, MFMA op counts
  rocprofv3 --kernel-trace --pmc $MFMA --output-format csv -d "$OUT/pmc_mfma_$mode" -- $B $F $P > "$OUT/pmc_mfma_$mode.log" 2>&1
done
# rows f1-f4 (heads, decode, backbone, whole network): bench lines, kernel stats, MFMA counters, launch order
python3 tools/heads_bench.py > "$OUT/heads_bench.json" 2>/dev/null
python3 tools/decode_bench.py > "$OUT/decode_bench.json" 2>/dev/null
python3 tools/backbone_bench.py > "$OUT/backbone_bench.json" 2>/dev/null
python3 tools/e2e_native_bench.py --graph > "$OUT/e2e_native.json" 2>/dev/null
python3 tools/generic_bench.py > "$OUT/generic_bench.json" 2>/dev/null
# round 6: overflow rate of the serving schedule on fresh batches per calibration policy (the AP50-delta proxy, ~8 min with
# its CPU side, is run on its own: python tests/proxy_ap.py --images 256 --out gpurun_out/$R/proxy_ap.json)
python3 tools/serving_margin_sweep.py > "$OUT/serving_margin_sweep.json" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/e2e_stats" -- python3 tools/e2e_native_bench.py > "$OUT/e2e_stats.log" 2>&1
rocprofv3 --kernel-trace --pmc $MFMA --output-format csv -d "$OUT/pmc_mfma_e2e" -- python3 tools/e2e_native_bench.py --steps 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_stats" -- python3 tools/train_step_bench.py > "$OUT/train_stats.log" 2>&1
rocprofv3 --kernel-trace --output-format csv -d "$OUT/bb_trace" -- python3 tools/backbone_bench.py > "$OUT/bb_trace.log" 2>&1
f=$(find "$OUT/bb_trace" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_order.py "$f" stem_kernel > "$OUT/backbone_kernel_order.txt"
# frozen serving mode end to end (byte codes from the stem to the heads' outputs / stages only): steady-state kernel
# time per batch, computed here from the trace (the traces themselves are large)
for mode in bytes stages_only; do
  arg=""; [ $mode = stages_only ] && arg=stages_only
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/fz_$mode" -o fz -- python3 tools/prof_e2e_frozen.py $arg > "$OUT/fz_$mode.log" 2>&1
  f=$(find "$OUT/fz_$mode" -name "fz_kernel_trace.csv" | head -1)
  m=stemq8_kernel; [ $mode = stages_only ] && m="stem_kernel<"
  [ -n "$f" ] && python3 tools/steady_stats.py "$f" "marker:$m" erfinv > "$OUT/e2e_frozen_${mode}_steady.txt"
  rm -rf "$OUT/fz_$mode"
done
for n in e2e_stats train_stats stats stats_frozen; do
  f=$(find "$OUT/$n" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${n}_kernel_stats.csv"
done
# the per-dispatch traces are large and not needed once the stats are extracted
find "$OUT" -name "*kernel_trace.csv" -path "*stats*" -delete
find "$OUT/bb_trace" -name "*.csv" -delete
ls "$OUT"
