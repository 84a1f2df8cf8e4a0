#!/bin/bash
# A/B of two builds on one box: lib/ (working tree) against lib_old/ (a build of another commit), alternating.  usage: tools/ab_lib.sh [repeats]
n=${1:-2}
for i in $(seq $n); do
  for so in "" $PWD/dynamic-visual-slam_amd/lib_old/libdvslam_hip.so; do
    DVSLAM_HIP_SO=$so python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${so:+old}' or 'new', d['value'], d['ms_per_step'], 'isolated', d['stage_ms_per_launch_isolated'], 'overlapped', d['stage_ms_per_launch_overlapped'])"
  done
done
