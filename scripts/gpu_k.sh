set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -30 > gpurun_out/k.log
rc=$?
cat gpurun_out/k.log
exit $rc
