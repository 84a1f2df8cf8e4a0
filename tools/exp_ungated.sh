mkdir -p gpurun_out/ug
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/ug/$name.json 2> gpurun_out/ug/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/ug/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['match_check'])" || tail -3 gpurun_out/ug/$name.err; }
run base X=1
run blur_ug DVS_BLUR_UNGATED=1
run match_ug BENCH_MATCH_UNGATED=1
run match_ug_lo BENCH_MATCH_UNGATED=1 BENCH_M_PRIO=-1
run both_ug DVS_BLUR_UNGATED=1 BENCH_MATCH_UNGATED=1 BENCH_M_PRIO=-1
run base2 X=1
