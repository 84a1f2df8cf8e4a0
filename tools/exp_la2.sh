mkdir -p gpurun_out/la2
run() { name=$1; shift; envs=""; while [ "$1" != "--" ] && [ -n "$1" ]; do envs="$envs $1"; shift; done; shift; env $envs timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/la2/$name.json 2> gpurun_out/la2/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/la2/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['match_check'], d['stage_ms_per_launch_overlapped'])" || tail -3 gpurun_out/la2/$name.err; }
run base X=1 --
run la DVS_LOOKAHEAD=1 --
run la_nodefer DVS_LOOKAHEAD=1 -- --defer off
run la_prio DVS_LOOKAHEAD=1 DVS_FA_PRIO=1 --
