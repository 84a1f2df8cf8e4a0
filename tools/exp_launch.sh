mkdir -p gpurun_out/ml
runl() { name=$1; shift; timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/ml/$name.json 2> gpurun_out/ml/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/ml/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/ml/$name.err; }
runl launch_default
runl launch_serial --serial-match
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ml/plain.json 2>gpurun_out/ml/plain.err; python -c "
import json
d=json.loads(open('gpurun_out/ml/plain.json').read().strip().splitlines()[-1]); print('plain', d['value'], d['ms_per_step'])"
