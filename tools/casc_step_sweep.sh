#!/bin/bash
# Run ON THE GPU BOX: whole-step rate (bench.py) for the pyramid cascade's tile sizes and for the seven-launch chain instead (DVS_CASCADE=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/lb; mkdir -p $OUT
run() { # B, env...
  B=$1; shift
  v=$(env "$@" python3 $R/bench.py --batch $B --steps 300 --warmup 20 --no-cpu-baseline --resident-batches 4 2>>$OUT/err2.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("oracle_check"))')
  echo "B=$B $* -> $v" | tee -a $OUT/casc_step.log
}
for B in 1 2 4 8; do
  run $B DVS_CASC_TW=128 DVS_CASC_TH=64
  run $B DVS_CASC_TW=64 DVS_CASC_TH=16
  run $B DVS_CASC_TW=128 DVS_CASC_TH=32
  run $B DVS_CASCADE=0
done
