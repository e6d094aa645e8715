set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "partitioned" 2>&1 | tail -30 > gpurun_out/pp.log
rc=$?
cat gpurun_out/pp.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python scripts/gpu_part_sim.py 2 > gpurun_out/part_sim2.log 2>&1 || { tail gpurun_out/part_sim2.log; exit 1; }
head -2 gpurun_out/part_sim2.log | cut -c1-300; tail -1 gpurun_out/part_sim2.log
timeout -k 10 500 python scripts/gpu_part_sim.py 4 > gpurun_out/part_sim4.log 2>&1 || { tail gpurun_out/part_sim4.log; exit 1; }
tail -1 gpurun_out/part_sim4.log
