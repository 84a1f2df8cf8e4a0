mkdir -p gpurun_out/ml
run() { name=$1; shift; timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/ml/$name.json 2> gpurun_out/ml/$name.err; python -c "
import json
d=json.loads(open('gpurun_out/ml/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], d['match_check'], d['config']['matches_lt50_frame1'])" || tail -3 gpurun_out/ml/$name.err; }
run default
run serial --serial-match
run b8 --batch 8
run b8serial --batch 8 --serial-match
run b1 --batch 1
run b1serial --batch 1 --serial-match
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ml/launch1.json 2> gpurun_out/ml/launch1.err; python -c "
import json
d=json.loads(open('gpurun_out/ml/launch1.json').read().strip().splitlines()[-1]); print('launch1', d['value'], d['ms_per_step'], d['rccl'])" || tail -5 gpurun_out/ml/launch1.err
