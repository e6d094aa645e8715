set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
p() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), d['config']['blocks'], d.get('sweep'), {k: round(v,2) for k,v in d['stages_ms_per_step'].items()})"; }
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/ov1.log 2>&1 || { tail -20 gpurun_out/ov1.log; exit 1; }
p < gpurun_out/ov1.log
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --serial-sweep > gpurun_out/ov2.log 2>&1 || { tail -20 gpurun_out/ov2.log; exit 1; }
p < gpurun_out/ov2.log
run() { timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --no-cpu-baseline "$@" > gpurun_out/mr.log 2>&1 || { tail -30 gpurun_out/mr.log; exit 1; }; grep '^{' gpurun_out/mr.log | p; }
run --cols-per-gpu 300000
run --cols-per-gpu 300000 --serial-sweep
run --cols-per-gpu 300000 --replicated-index
run --cols-per-gpu 300000 --force-row-pairs 500
