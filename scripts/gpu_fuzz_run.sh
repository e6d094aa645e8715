set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python scripts/gpu_fuzz.py ${1:-600} ${2:-777000} > gpurun_out/fuzz.log 2>&1
rc=$?
tail -3 gpurun_out/fuzz.log
exit $rc
