mkdir -p gpurun_out/sweep
for b in 1 2 4 8 16 32 64 128; do timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --batch $b > gpurun_out/sweep/b$b.json 2>gpurun_out/sweep/b$b.err; python -c "
import json; d=json.loads(open('gpurun_out/sweep/b$b.json').read().strip().splitlines()[-1]); print($b, d['value'], d['ms_per_step'])"; done
timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --serial-match > gpurun_out/sweep/serial.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/sweep/serial.json').read().strip().splitlines()[-1]); print('serial', d['value'], d['ms_per_step'])"
timeout -k 10 200 python tools/latency.py 2>&1 | tail -4
