# round 4, second session, last collection: GPU suite, profile set, bench line
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests_final.log 2>&1 || { tail -40 gpurun_out/r04_gputests_final.log; exit 1; }
tail -2 gpurun_out/r04_gputests_final.log
timeout -k 10 300 python scripts/gpu_fuzz.py 180 1500000 > gpurun_out/r04_fuzz_final2.log 2>&1 || { tail -20 gpurun_out/r04_fuzz_final2.log; exit 1; }
tail -1 gpurun_out/r04_fuzz_final2.log
bash scripts/gpu_r4_final_prof.sh
