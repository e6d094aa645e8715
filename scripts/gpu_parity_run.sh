set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/parity.log
rc=$?
cat gpurun_out/parity.log
exit $rc
