mkdir -p gpurun_out/r2h
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > gpurun_out/r2h/$name.json 2> gpurun_out/r2h/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/r2h/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['stage_ms_per_launch_overlapped'])"; }
run base A=1
run nosplit DVS_DESC_SPLIT=0
run oct512 DVS_OCT_T=512
run pfafter DVS_PF_AFTER_FAST=1
run pfafter_nosplit DVS_PF_AFTER_FAST=1 DVS_DESC_SPLIT=0
run all3 DVS_PF_AFTER_FAST=1 DVS_DESC_SPLIT=0 DVS_OCT_T=512
