mkdir -p gpurun_out/sw
run() { name=$1; shift; envs=""; while [ "$1" != "--" ] && [ -n "$1" ]; do envs="$envs $1"; shift; done; shift; env $envs timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/sw/$name.json 2> gpurun_out/sw/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/sw/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/sw/$name.err; }
run base X=1 --
run oct512 DVS_OCT_T=512 --
run auxhi DVS_AUX_PRIO=1 --
run nodefer X=1 -- --defer off
run pflo DVS_PF_PRIO=-1 --
run mainhi DVS_MAIN_PRIO=1 --
run base2 X=1 --
run b128 X=1 -- --batch 128
run b32 X=1 -- --batch 32
