set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python scripts/gpu_part_sim.py ${1:-4} ${2:-1000000} > gpurun_out/part_sim.log 2>&1 || { tail -30 gpurun_out/part_sim.log; exit 1; }
cat gpurun_out/part_sim.log
