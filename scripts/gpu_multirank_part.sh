set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for extra in "" "--replicated-index"; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --cols-per-gpu 500000 --no-cpu-baseline $extra > gpurun_out/mr2part.log 2>&1 || { tail -30 gpurun_out/mr2part.log; exit 1; }
tail -1 gpurun_out/mr2part.log | cut -c1-1500
done
