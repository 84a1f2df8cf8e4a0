#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel statistics + PMC passes of `bench.py`, one counter group per pass
# (never --pmc together with a trace domain other than --kernel-trace).  Everything lands in gpurun_out/<tag>/; summarise
# afterwards with tools/pmc_summary.py and copy what should be judged into profiles/.
#   usage: tools/collect_profiles.sh <tag> [steps]
set -e -o pipefail
TAG=${1:-prof}; STEPS=${2:-5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, extra env assignment ("" for none), rocprofv3 args...
  local name=$1; shift
  echo "== $name" >> $OUT/progress.log
  timeout -k 10 240 rocprofv3 "$@" -d $OUT/$name -o $name -- python3 $R/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline > $OUT/$name.log 2>&1
}
# 1. kernel statistics of the real (overlapped) run, then with every stage alone on the stream
run stats_overlap --kernel-trace --stats --output-format csv
export DVS_NO_OVERLAP=1
run stats_isolated --kernel-trace --stats --output-format csv
# 2. PMC passes (isolated launches so a counter belongs to one kernel)
run pmc_fetch --pmc FETCH_SIZE --kernel-trace
run pmc_write --pmc WRITE_SIZE --kernel-trace
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace
run pmc_insts --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace
run pmc_busy --pmc VALUBusy SALUBusy --kernel-trace
run pmc_mem --pmc MemUnitBusy LDSBankConflict --kernel-trace
echo done >> $OUT/progress.log
