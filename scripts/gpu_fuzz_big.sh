set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FBG_FUZZ_BIG=1 timeout -k 10 1000 python scripts/gpu_fuzz.py ${1:-600} ${2:-5000} > gpurun_out/fuzz_big.log 2>&1
rc=$?
tail -3 gpurun_out/fuzz_big.log
exit $rc
