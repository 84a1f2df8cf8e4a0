# FAST-only PMC: wave cycles split into parked / issue-stalled / active, with and without byte-aligned DMA
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/fastpmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DVS_NO_OVERLAP=1
for mode in 1 0; do
  export DVS_FAST_BYTE_DMA=$mode
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/a$mode -o a -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/a$mode.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d $OUT/b$mode -o b -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/b$mode.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SALU --kernel-trace -d $OUT/c$mode -o c -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/c$mode.log 2>&1 || exit 1
  echo "mode $mode done" >> $OUT/progress.log
done
