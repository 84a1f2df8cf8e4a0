mkdir -p gpurun_out/host
run() { name=$1; shift; timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline "$@" > gpurun_out/host/$name.json 2> gpurun_out/host/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/host/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['match_check'], d['host_enqueue_ms_per_step'])" || tail -3 gpurun_out/host/$name.err; }
run b1 --batch 1
run b2 --batch 2
run b4 --batch 4
run b8 --batch 8
run b64
run b64serial --serial-match
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/host/l1.json 2> gpurun_out/host/l1.err; python -c "
import json
d=json.loads(open('gpurun_out/host/l1.json').read().strip().splitlines()[-1]); print('launch1', d['value'], d['ms_per_step'], d['match_check'], d['host_enqueue_ms_per_step'])" || tail -5 gpurun_out/host/l1.err
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 300 --warmup 30 --no-cpu-baseline --batch 1 > gpurun_out/host/l1b1.json 2> gpurun_out/host/l1b1.err; python -c "
import json
d=json.loads(open('gpurun_out/host/l1b1.json').read().strip().splitlines()[-1]); print('launch1 b1', d['value'], d['ms_per_step'], d['match_check'], d['host_enqueue_ms_per_step'])" || tail -5 gpurun_out/host/l1b1.err
