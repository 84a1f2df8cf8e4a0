mkdir -p gpurun_out/r2n
run() { name=$1; shift; flags=""; if [ "$1" = "--" ]; then shift; flags="$1"; shift; fi; env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline $flags > gpurun_out/r2n/$name.json 2> gpurun_out/r2n/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/r2n/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['stage_ms_per_launch_overlapped'])"; }
run la0_serial DVS_LOOKAHEAD=0
run la1_serial DVS_LOOKAHEAD=1
run la1_serial_norm DVS_LOOKAHEAD=1 DVS_FA_PRIO=0
run la1_serial_hi DVS_LOOKAHEAD=1 DVS_FA_PRIO=1
run la0_ms -- --match-stream DVS_LOOKAHEAD=0
run la1_ms -- --match-stream DVS_LOOKAHEAD=1
