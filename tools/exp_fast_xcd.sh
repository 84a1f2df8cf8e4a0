set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/fxcd; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DVS_NO_OVERLAP=1
for mode in 1 0; do
  export DVS_FAST_XCD=$mode
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/f$mode -o f -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/f$mode.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/t$mode -o t -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/t$mode.log 2>&1 || exit 1
done
unset DVS_NO_OVERLAP
cd $R
for mode in 1 0 1 0; do DVS_FAST_XCD=$mode timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/b$mode.json 2>$OUT/b$mode.err; python -c "
import json; d=json.loads(open('$OUT/b$mode.json').read().strip().splitlines()[-1]); print('xcd=$mode', d['value'], d['ms_per_step'], d['stage_ms_per_launch_isolated']['fast'])"; done
