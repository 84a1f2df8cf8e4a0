mkdir -p gpurun_out/flags
run() { name=$1; shift; timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/flags/$name.json 2> gpurun_out/flags/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/flags/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['match_check'])" || tail -3 gpurun_out/flags/$name.err; }
run default
run serial --serial-match
run noprefetch --no-prefetch
run nodefer --defer off
run mstream --match-stream --serial-match
run single --single-resident-batch
run serial_noprefetch --serial-match --no-prefetch
run b3 --batch 3
run levels --shard levels --batch 4
run torchx --torch-exchange
