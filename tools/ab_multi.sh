#!/bin/bash
# A/B of several builds of the library in ONE call on ONE box (boxes differ by a few per cent):
#   bash tools/ab_multi.sh [-r ROUNDS] codenet_amd/lib/libcodenet_dcn.so codenet_amd/lib/libcodenet_dcn_<tag>.so ...
# Rounds interleave the builds; per run: ms per step (running schedule, median region), per-launch kernel times (us),
# ms per step (frozen schedule).
R=2
[ "$1" = "-r" ] && { R=$2; shift 2; }
for i in $(seq $R); do
  for V in "$@"; do
    python3 tools/with_lib.py "$V" bench.py --no-cpu-baseline --no-e2e --no-config-legs 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms_per_launch']
print('%-34s %.4f  min %.4f | %s | frozen %.4f' % ('$V'.split('/')[-1], d['ms_per_step'], d['regions']['ms_per_step_min'], ' '.join('%s %.1f' % (n.replace('pointwise','pw').replace('scale','sc'), v*1e3) for n,v in k.items() if 'unpack' not in n), d['frozen_int8']['ms_per_step']))"
  done
done
