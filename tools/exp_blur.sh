mkdir -p gpurun_out/blur
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/blur/$name.json 2> gpurun_out/blur/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/blur/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], 'blur iso', d['stage_ms_per_launch_isolated']['blur'], 'ovl', d['stage_ms_per_launch_overlapped']['blur'])"; }
run base DVS_BLUR_DBG=0
run nostore DVS_BLUR_DBG=1
run samerow DVS_BLUR_DBG=2
run both DVS_BLUR_DBG=3
