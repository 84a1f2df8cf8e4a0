mkdir -p gpurun_out/rep
for r in 1 2 3; do timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline "$@" > gpurun_out/rep/r$r.json 2> gpurun_out/rep/r$r.err; python -c "
import json
d=json.loads(open('gpurun_out/rep/r$r.json').read().strip().splitlines()[-1]); print('rep$r', d['value'], d['ms_per_step'], d['match_check'], d['stage_ms_per_launch_isolated']['octree'], d['stage_ms_per_launch_overlapped'])"; done
