mkdir -p gpurun_out/la
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/la/$name.json 2> gpurun_out/la/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/la/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['stage_ms_per_launch_overlapped'])" || tail -3 gpurun_out/la/$name.err; }
run prio_base DVS_LOOKAHEAD=0
run prio_la DVS_LOOKAHEAD=1
run prio_la_hi DVS_LOOKAHEAD=1 DVS_FA_PRIO=1
run prio_ms DVS_LOOKAHEAD=0 BENCH_X=1
