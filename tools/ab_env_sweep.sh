#!/bin/bash
# Run ON THE GPU BOX: bench.py at batch sizes under two environments, alternating (same box), value and ms per step
#   usage: tools/ab_env_sweep.sh <tag> "<B list>" "<ENV=a>" "<ENV=b>" [rounds]
TAG=$1; BS=$2; EA=$3; EB=$4; ROUNDS=${5:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for B in $BS; do
  for r in $(seq $ROUNDS); do
    for E in "$EA" "$EB"; do
      v=$(env $E python3 $R/bench.py --batch $B --steps 300 --warmup 20 --no-cpu-baseline --resident-batches 4 2>>$OUT/err.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d.get("match_check"))')
      echo "B=$B $E -> $v" | tee -a $OUT/ab.log
    done
  done
done
