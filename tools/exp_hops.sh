mkdir -p gpurun_out/hops
run() { name=$1; shift; envs=""; while [ "$1" != "--" ] && [ -n "$1" ]; do envs="$envs $1"; shift; done; shift; env $envs timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline "$@" > gpurun_out/hops/$name.json 2> gpurun_out/hops/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/hops/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['match_check'])" || tail -3 gpurun_out/hops/$name.err; }
for r in 1 2; do
run b64_auto_$r X=1 -- 
run b64_defer_$r X=1 -- --defer on
run b48_defer_$r X=1 -- --defer on --batch 48
run b48_nodefer_$r X=1 -- --defer off --batch 48
run b128_nodefer_$r X=1 -- --defer off --batch 128
run b128_defer_$r X=1 -- --defer on --batch 128
done
