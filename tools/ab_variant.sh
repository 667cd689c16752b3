#!/bin/bash
# A/B of a variant build of the library against the committed one on the GPU box:
#   bash tools/ab_variant.sh codenet_amd/lib/libcodenet_dcn_<tag>.so [pytest files...]
# prints ms per step (running schedule), ms inside the gather kernels, ms per step (frozen schedule), twice each,
# then runs the given parity tests against the variant.
V="$GRAFT_REPO_ROOT/$1"; shift
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['roofline']['ms_per_step_in_kernel'],4), round(d['frozen_int8']['ms_per_step'],4))"; }
for i in 1 2; do
  python3 bench.py --no-cpu-baseline --no-e2e 2>/dev/null | show base
  python3 tools/with_lib.py "$V" bench.py --no-cpu-baseline --no-e2e 2>/dev/null | show variant
done
[ $# -gt 0 ] && python3 tools/with_lib.py "$V" -m pytest "$@" -m gpu -q 2>&1 | tail -2
