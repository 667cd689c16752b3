#!/bin/bash
# On the GPU box: kernel stats of the frozen serving network, byte-code backbone and stages-only (csv under gpurun_out/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in bytes stages_only; do
  arg=""; [ $mode = stages_only ] && arg=stages_only
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fz_$mode -o fz -- python3 tools/prof_e2e_frozen.py $arg > gpurun_out/prof_fz_$mode.log 2>&1
done
find gpurun_out -name "fz_kernel_stats.csv"
