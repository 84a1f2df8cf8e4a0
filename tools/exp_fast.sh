mkdir -p gpurun_out/fast
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/fast/$name.json 2> gpurun_out/fast/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/fast/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], 'fast iso', d['stage_ms_per_launch_isolated']['fast'], 'ovl', d['stage_ms_per_launch_overlapped']['fast'])"; }
run aligned DVS_FAST_BYTE_DMA=0
run bytedma DVS_FAST_BYTE_DMA=1
run aligned2 DVS_FAST_BYTE_DMA=0
run bytedma2 DVS_FAST_BYTE_DMA=1
