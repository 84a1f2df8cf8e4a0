#!/bin/bash
# A/B/... of (library, environment) pairs on one box, alternating: tools/ab_mixed.sh <repeats> "<so or -> [NAME=V ...]" ...   (BENCH_ARGS: extra bench.py flags)
n=$1; shift
for i in $(seq $n); do
  for spec in "$@"; do
    so=$(echo $spec | cut -d' ' -f1); envs=$(echo $spec | cut -s -d' ' -f2-)
    p=$so; [ "$so" = "-" ] && p=""
    env DVSLAM_HIP_SO=$p $envs python3 bench.py --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$spec]', d['value'], d['ms_per_step'], 'overlapped', d['stage_ms_per_launch_overlapped'])"
  done
done
