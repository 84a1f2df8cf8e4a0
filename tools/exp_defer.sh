mkdir -p gpurun_out/df
run() { name=$1; shift; timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > gpurun_out/df/$name.json 2> gpurun_out/df/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/df/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/df/$name.err; }
for b in 1 4 16 32 64; do run d$b --batch $b; run n$b --batch $b --no-defer; done
