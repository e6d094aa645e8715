set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/bp_tests.log
rc=$?
cat gpurun_out/bp_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/gpu_configs.py c5 star > gpurun_out/bp_perf.log 2>&1
rc=$?
cat gpurun_out/bp_perf.log
exit $rc
