mkdir -p gpurun_out/occ
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/occ/$name.json 2> gpurun_out/occ/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/occ/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], 'fast iso', d['stage_ms_per_launch_isolated']['fast'], 'ovl', d['stage_ms_per_launch_overlapped']['fast'], d['config']['keypoints_frame1'])" || tail -3 gpurun_out/occ/$name.err; }
run base X=1
run list1536 DVS_FAST_LIST=1536
run list1024 DVS_FAST_LIST=1024
run base2 X=1
