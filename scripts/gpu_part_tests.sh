set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "partitioned" 2>&1 | tail -40 > gpurun_out/part.log
rc=$?
cat gpurun_out/part.log
exit $rc
