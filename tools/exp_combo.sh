mkdir -p gpurun_out/occ
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/occ/$name.json 2> gpurun_out/occ/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/occ/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/occ/$name.err; }
for r in 1 2 3; do
run t1s1_$r DVS_PYR_TRIPLE=1 DVS_FAST_SPLIT=1
run t1s0_$r DVS_PYR_TRIPLE=1 DVS_FAST_SPLIT=0
run t0s1_$r DVS_PYR_TRIPLE=0 DVS_FAST_SPLIT=1
run t0s0_$r DVS_PYR_TRIPLE=0 DVS_FAST_SPLIT=0
done
