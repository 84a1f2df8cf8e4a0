#!/bin/bash
# experiment driver: lines of "<label> <so or -> <ENV=VAL ...>" on stdin, alternating $1 times
n=$1
mapfile -t rows
for i in $(seq $n); do
  for r in "${rows[@]}"; do
    set -- $r; label=$1; so=$2; shift 2
    p=$so; [ "$so" = "-" ] && p=""
    env "$@" DVSLAM_HIP_SO=$p python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
o=d['stage_ms_per_launch_overlapped']; a=d['stage_ms_per_launch_isolated']
print('$label', d['value'], d['ms_per_step'], 'fast', a['fast'], 'ovl: pyr', o['pyramid'], 'fast', o['fast'], 'oct', o['octree'], 'blur', o['blur'], 'desc', o['describe'])"
  done
done
