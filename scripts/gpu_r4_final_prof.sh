# round 4, second session: the profile set of scripts/gpu_r4_prof.sh on the final code, then the bench line
set -x
bash $GRAFT_REPO_ROOT/scripts/gpu_r4_prof.sh || exit 1
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err || { tail -20 gpurun_out/r04_bench.err; exit 1; }
tail -c 600 gpurun_out/r04_bench.json
