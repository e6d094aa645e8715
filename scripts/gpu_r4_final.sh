# round 4, second session: the whole GPU suite and the two fuzz campaigns on the final code
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests_final.log 2>&1 || { tail -40 gpurun_out/r04_gputests_final.log; exit 1; }
tail -2 gpurun_out/r04_gputests_final.log
timeout -k 10 400 python scripts/gpu_fuzz.py 240 1700000 > gpurun_out/r04_fuzz_final.log 2>&1 || { tail -20 gpurun_out/r04_fuzz_final.log; exit 1; }
tail -1 gpurun_out/r04_fuzz_final.log
timeout -k 10 300 python scripts/gpu_fuzz_dp.py 180 1800000 > gpurun_out/r04_fuzz_dp_final.log 2>&1 || { tail -20 gpurun_out/r04_fuzz_dp_final.log; exit 1; }
tail -1 gpurun_out/r04_fuzz_dp_final.log
