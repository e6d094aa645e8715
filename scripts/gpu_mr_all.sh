set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 1 --backend gloo --no-cpu-baseline "$@" > gpurun_out/mr.log 2>&1 || { tail -30 gpurun_out/mr.log; exit 1; }; tail -1 gpurun_out/mr.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), d['config']['blocks'], d['config']['workload'][-90:])"; }
run --cols-per-gpu 300000
run --cols-per-gpu 300000 --replicated-index
run --cols-per-gpu 300000 --force-row-pairs 500
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), d['config']['blocks'])"
timeout -k 10 600 python scripts/gpu_part_sim.py 8 1000000 --sequential > gpurun_out/part_sim8.log 2>&1 || { tail gpurun_out/part_sim8.log; exit 1; }
tail -2 gpurun_out/part_sim8.log
