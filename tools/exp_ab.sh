# usage: exp_ab.sh VAR  -> ABAB of VAR=1 / VAR=0 with the default bench
mkdir -p gpurun_out/ab
for r in 1 2 3 4; do for mode in 1 0; do env $1=$mode timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/ab/b.json 2>gpurun_out/ab/b.err; python -c "
import json; d=json.loads(open('gpurun_out/ab/b.json').read().strip().splitlines()[-1]); print('$1=$mode', d['value'], d['ms_per_step'])"; done; done
