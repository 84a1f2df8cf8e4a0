#!/bin/bash
# A/B/... of several builds on one box, alternating: tools/ab_libs.sh <repeats> <so> [<so> ...]  ("-" = the in-tree lib/)
n=$1; shift
for i in $(seq $n); do
  for so in "$@"; do
    p=$so; [ "$so" = "-" ] && p=""
    DVSLAM_HIP_SO=$p python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$so', d['value'], d['ms_per_step'], 'isolated', d['stage_ms_per_launch_isolated'], 'overlapped', d['stage_ms_per_launch_overlapped'])"
  done
done
