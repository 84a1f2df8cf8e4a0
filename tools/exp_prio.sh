mkdir -p gpurun_out/prio
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/prio/$name.json 2> gpurun_out/prio/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/prio/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['stage_ms_per_launch_overlapped'])" || tail -3 gpurun_out/prio/$name.err; grep "ten latest" gpurun_out/prio/$name.err | head -1; }
run auxlo DVS_DEBUG=6 DVS_AUX_PRIO=-1
run auxlo_mlo DVS_DEBUG=6 DVS_AUX_PRIO=-1 BENCH_M_PRIO=-1
run auxlo_mainhi DVS_DEBUG=6 DVS_AUX_PRIO=-1 DVS_MAIN_PRIO=1
run auxlo_mainhi_mlo DVS_DEBUG=6 DVS_AUX_PRIO=-1 DVS_MAIN_PRIO=1 BENCH_M_PRIO=-1
run aux0_mainhi DVS_DEBUG=6 DVS_AUX_PRIO=0 DVS_MAIN_PRIO=1
run auxlo_mhi DVS_DEBUG=6 DVS_AUX_PRIO=-1 BENCH_M_PRIO=1
