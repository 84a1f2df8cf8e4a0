mkdir -p gpurun_out/prio
for b in 1 4 8 16 32 64 128; do for p in 1 -1; do DVS_AUX_PRIO=$p timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --batch $b > gpurun_out/prio/s.json 2>gpurun_out/prio/s.err; python -c "
import json; d=json.loads(open('gpurun_out/prio/s.json').read().strip().splitlines()[-1]); print('batch $b aux_prio $p', d['value'], d['ms_per_step'])"; done; done
