# round 4, second session: the whole GPU suite, the general fuzz campaign (5 min) and the sweep's (3 min) on the final code
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04b_gputests.log 2>&1 || { tail -40 gpurun_out/r04b_gputests.log; exit 1; }
tail -2 gpurun_out/r04b_gputests.log
timeout -k 10 400 python scripts/gpu_fuzz.py 300 1200000 > gpurun_out/r04b_fuzz.log 2>&1 || { tail -20 gpurun_out/r04b_fuzz.log; exit 1; }
tail -1 gpurun_out/r04b_fuzz.log
timeout -k 10 300 python scripts/gpu_fuzz_dp.py 180 1300000 > gpurun_out/r04b_fuzz_dp.log 2>&1 || { tail -20 gpurun_out/r04b_fuzz_dp.log; exit 1; }
tail -1 gpurun_out/r04b_fuzz_dp.log
