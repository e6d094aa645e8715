set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "wide or FORCE_WIDE or beyond_32bit" 2>&1 | tail -40 > gpurun_out/wide.log
rc=$?
cat gpurun_out/wide.log
exit $rc
