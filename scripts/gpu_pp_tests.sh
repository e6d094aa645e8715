set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q -k "partitioned" 2>&1 | tail -15 > gpurun_out/pp.log
rc=$?
cat gpurun_out/pp.log
[ $rc -eq 0 ] || exit $rc
bash scripts/gpu_prof_part.sh 8 1000000 --sequential
f=$(find gpurun_out/prof_part -name "*kernel_stats.csv" | head -1); grep -h "k_pp_" $f | cut -d, -f1-4
