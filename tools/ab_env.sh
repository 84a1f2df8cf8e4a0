#!/bin/bash
# A/B of one environment switch on one box: tools/ab_env.sh NAME VALUE_A VALUE_B [repeats]  (bench.py --no-cpu-baseline, alternating)
name=$1; a=$2; b=$3; n=${4:-2}
for i in $(seq $n); do
  for v in $a $b; do
    env $name=$v python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name=$v', d['value'], d['ms_per_step'], 'isolated', d['stage_ms_per_launch_isolated'], 'overlapped', d['stage_ms_per_launch_overlapped'])"
  done
done
