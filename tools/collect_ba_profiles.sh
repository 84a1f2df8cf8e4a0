#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel statistics of the BA evaluation (single window, batched) and the device LM solve.
set -e -o pipefail
TAG=${1:-ba}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in ba_prof ba_prof_batched ba_lm_prof; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$t -o $t -- python3 $R/tools/$t.py > $OUT/$t.log 2>&1
  echo "== $t" >> $OUT/progress.log
done
