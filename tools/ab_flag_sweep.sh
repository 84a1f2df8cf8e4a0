#!/bin/bash
# Run ON THE GPU BOX: bench.py at batch sizes with two sets of extra flags, alternating (same box)
#   usage: tools/ab_flag_sweep.sh <tag> "<B list>" "<flags a>" "<flags b>" [rounds]
TAG=$1; BS=$2; FA=$3; FB=$4; ROUNDS=${5:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for B in $BS; do
  for r in $(seq $ROUNDS); do
    for F in "$FA" "$FB"; do
      v=$(python3 $R/bench.py --batch $B --steps 300 --warmup 20 --no-cpu-baseline --resident-batches 4 $F 2>>$OUT/err.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d.get("match_check"))')
      echo "B=$B [$F] -> $v" | tee -a $OUT/ab.log
    done
  done
done
