for r in 1 2; do timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/rep/x.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/rep/x.json').read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], 'blur iso', d['stage_ms_per_launch_isolated']['blur'], 'ovl', d['stage_ms_per_launch_overlapped']['blur'])"; done
