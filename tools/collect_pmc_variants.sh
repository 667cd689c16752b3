#!/bin/bash
# Runs on the GPU box (via gpurun): PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes, --kernel-trace only) for the
# variants the bench and the README quote besides the default schedules (VERDICT r3 missing #3 / "next" #8): cfg4, cfg2,
# the byte-code schedule with byte-code input + chained scale, the whole network in serving mode (e2e.frozen), the QAT
# step.  Output: gpurun_out/<round>/pmc_traffic_<variant>.json (tools/pmc_steady.py: bytes per steady-state iteration).
set -u
R=${1:-r06}
OUT="gpurun_out/$R"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P="--steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-e2e --no-config-legs"
run() {   # name, marker, per-iter, command...
  local name=$1 marker=$2 per=$3; shift 3
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pv_${name}_f" -- "$@" > "$OUT/pv_${name}.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pv_${name}_w" -- "$@" >> "$OUT/pv_${name}.log" 2>&1
  python3 tools/pmc_steady.py "$OUT/pv_${name}_f" "$OUT/pv_${name}_w" "$marker" --per-iter "$per" > "$OUT/pmc_steady_${name}.json" 2>> "$OUT/pv_${name}.log"
  rm -rf "$OUT/pv_${name}_f" "$OUT/pv_${name}_w"
  echo "$name: $(cut -c1-300 "$OUT/pmc_steady_${name}.json")"
}
run cfg3 scale_nchw_kernel 1 python3 bench.py $P
run cfg4 scale_nchw_kernel 1 python3 bench.py --config cfg4 $P
run cfg2 scale_nchw_kernel 1 python3 bench.py --config cfg2 $P
run frozen scale_nchw_kernel 1 python3 bench.py --frozen $P
run frozen_chained frozen_params_kernel 1 python3 tools/prof_frozen_chained.py
run e2e_frozen stemq8_kernel 1 python3 tools/prof_e2e_frozen.py
run e2e stem_kernel 1 python3 tools/e2e_native_bench.py --steps 4
run train_step scale_kernel 3 python3 tools/train_step_bench.py --steps 4
ls "$OUT"
