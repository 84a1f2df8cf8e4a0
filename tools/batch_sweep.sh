#!/bin/bash
# Run ON THE GPU BOX: bench.py over batch sizes (and lane counts) on one GPU -> gpurun_out/<tag>/batch_sweep.json
#   usage: tools/batch_sweep.sh <tag> "<B list>" "<lanes list>" [steps]
TAG=${1:-sweep}; BS=${2:-"1 2 4 8 16 32 64"}; LANES=${3:-0}; STEPS=${4:-300}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
: > $OUT/rows.jsonl
for B in $BS; do
  for LN in $LANES; do
    python3 $R/bench.py --batch $B --lanes $LN --steps $STEPS --warmup 20 --no-cpu-baseline --resident-batches 4 2>> $OUT/err.log | tail -1 >> $OUT/rows.jsonl
    echo "B=$B lanes=$LN done" >> $OUT/progress.log
  done
done
python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
rows = []
for line in open(out + "/rows.jsonl"):
    try:
        d = json.loads(line)
    except Exception:
        continue
    rows.append({"frames_per_step": d["config"]["frames_per_gpu_per_step"], "schedule": d["config"]["parallelism"].split(", ")[-1][:60],
                 "frames_per_s": d["value"], "ms_per_step": d["ms_per_step"], "host_enqueue_ms_per_step": d["host_enqueue_ms_per_step"],
                 "steps": d["steps"], "oracle_check": d.get("oracle_check"), "match_check": d.get("match_check")})
json.dump({"what": "python bench.py --batch B --lanes L --no-cpu-baseline on one MI355X: extract + match, 1280x720 / 2000 kp, frames resident in HBM",
           "rows": rows}, open(out + "/batch_sweep.json", "w"), indent=1)
for r in rows:
    print(r["frames_per_step"], r["schedule"][:40], r["frames_per_s"], r["ms_per_step"], r["host_enqueue_ms_per_step"], r["match_check"])
PY
