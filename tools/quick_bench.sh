#!/bin/bash
# On the GPU box: bench.py twice without the CPU / whole-network legs; prints ms per step (running, median of the timed
# regions), per-launch kernel times, ms per step (frozen).  Optional arguments: pytest files to run first.
[ $# -gt 0 ] && python3 -m pytest "$@" -m gpu -x -q 2>&1 | tail -8
for i in 1 2; do
  python3 bench.py --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['kernel_ms_per_launch'].items()}, 'frozen', round(d['frozen_int8']['ms_per_step'],4))"
done
