mkdir -p gpurun_out/occ
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/occ/$name.json 2> gpurun_out/occ/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/occ/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], 'fast ovl', d['stage_ms_per_launch_overlapped']['fast'])" || tail -3 gpurun_out/occ/$name.err; }
run base X=1
run nomatch BENCH_NO_MATCH=1
run nomatch_list1024 BENCH_NO_MATCH=1 DVS_FAST_LIST=1024
