set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "gapped" 2>&1 | tail -5 > gpurun_out/gapped_tests.log
rc=$?
cat gpurun_out/gapped_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/gpu_configs.py c5gapped c3gapped > gpurun_out/gapped_perf.log 2>&1
rc=$?
cat gpurun_out/gapped_perf.log
exit $rc
