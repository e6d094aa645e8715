# round 4, second session: the whole GPU suite and the fuzz campaigns on the final code
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests_final.log 2>&1 || { tail -40 gpurun_out/r04_gputests_final.log; exit 1; }
tail -2 gpurun_out/r04_gputests_final.log
timeout -k 10 400 python scripts/gpu_fuzz.py 180 2500000 > gpurun_out/r04_fuzz_final.log 2>&1 || { tail -20 gpurun_out/r04_fuzz_final.log; exit 1; }
tail -1 gpurun_out/r04_fuzz_final.log
timeout -k 10 300 python scripts/gpu_fuzz_span.py 180 2600000 > gpurun_out/r04_fuzz_span_final.log 2>&1 || { tail -20 gpurun_out/r04_fuzz_span_final.log; exit 1; }
tail -1 gpurun_out/r04_fuzz_span_final.log
timeout -k 10 300 python scripts/gpu_fuzz_dp.py 120 2700000 > gpurun_out/r04_fuzz_dp_final.log 2>&1 || { tail -20 gpurun_out/r04_fuzz_dp_final.log; exit 1; }
tail -1 gpurun_out/r04_fuzz_dp_final.log
