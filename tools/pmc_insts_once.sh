set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DVS_NO_OVERLAP=1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace -d $OUT/i -o i -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/i.log 2>&1
