#!/bin/bash
# re-collection of the default-workload passes of tools/collect_profiles.sh / collect_pmc_variants.sh (round 5: the
# first collection ran them with the `configs` legs of bench.py inside the profiled run)
set -u
R=r05; OUT="gpurun_out/$R"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py"; Q="--no-cpu-baseline --no-e2e --no-config-legs"
rm -rf "$OUT/stats" "$OUT/stats_frozen" "$OUT"/pmc_*_running "$OUT"/pmc_*_frozen
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $B $Q > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_frozen" -- $B --frozen $Q > "$OUT/stats_frozen.log" 2>&1
P="--steps 3 --warmup 2 --no-graph $Q"
MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
for mode in running frozen; do
  F=""; [ $mode = frozen ] && F="--frozen"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$mode" -- $B $F $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$mode" -- $B $F $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_sq_$mode" -- $B $F $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $MFMA --output-format csv -d "$OUT/pmc_mfma_$mode" -- $B $F $P > "$OUT/pmc_mfma_$mode.log" 2>&1
done
for n in stats stats_frozen; do
  f=$(find "$OUT/$n" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${n}_kernel_stats.csv"
done
find "$OUT" -name "*kernel_trace.csv" -path "*stats*" -delete
PV="--steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-e2e --no-config-legs"
for name in cfg3 frozen; do
  F=""; [ $name = frozen ] && F="--frozen"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pv_${name}_f" -- python3 bench.py $F $PV > "$OUT/pv_${name}.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pv_${name}_w" -- python3 bench.py $F $PV >> "$OUT/pv_${name}.log" 2>&1
  python3 tools/pmc_steady.py "$OUT/pv_${name}_f" "$OUT/pv_${name}_w" scale_nchw_kernel --per-iter 1 > "$OUT/pmc_steady_${name}.json" 2>> "$OUT/pv_${name}.log"
  rm -rf "$OUT/pv_${name}_f" "$OUT/pv_${name}_w"
  echo "$name: $(cut -c1-200 "$OUT/pmc_steady_${name}.json")"
done
