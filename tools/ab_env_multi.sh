#!/bin/bash
# Run ON THE GPU BOX: bench.py at batch sizes under several environments in turn (same box)
#   usage: tools/ab_env_multi.sh <tag> "<B list>" rounds ENV1 ENV2 ...
TAG=$1; BS=$2; ROUNDS=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for B in $BS; do
  for r in $(seq $ROUNDS); do
    for E in "$@"; do
      v=$(env $E python3 $R/bench.py --batch $B --steps 300 --warmup 20 --no-cpu-baseline --resident-batches 4 2>>$OUT/err.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d.get("match_check"))')
      echo "B=$B $E -> $v" | tee -a $OUT/ab.log
    done
  done
done
