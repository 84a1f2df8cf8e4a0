#!/bin/bash
# A/B/... of environment settings on one box, alternating: tools/ab_env_list.sh <repeats> "<NAME=V ...>" ["<NAME=V ...>" ...]   ("-" = no setting)
n=$1; shift
for i in $(seq $n); do
  for e in "$@"; do
    s=$e; [ "$e" = "-" ] && s=""
    env $s python3 bench.py --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$e]', d['value'], d['ms_per_step'], 'overlapped', d['stage_ms_per_launch_overlapped'])"
  done
done
