# round 4, second session: the campaigns that build large texts and the span scan's, on the final code
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
FBG_FUZZ_BIG=1 timeout -k 10 500 python scripts/gpu_fuzz.py 360 2100000 > gpurun_out/r04_fuzz_big_final.log 2>&1 || { tail -20 gpurun_out/r04_fuzz_big_final.log; exit 1; }
tail -1 gpurun_out/r04_fuzz_big_final.log
timeout -k 10 400 python scripts/gpu_fuzz_span.py 240 2200000 > gpurun_out/r04_fuzz_span_final.log 2>&1 || { tail -20 gpurun_out/r04_fuzz_span_final.log; exit 1; }
tail -1 gpurun_out/r04_fuzz_span_final.log
