# round 3: the whole GPU suite, then the bench line (with extras) and its kernel stats
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3c_all.log 2>&1 || { tail -40 gpurun_out/r3c_all.log; exit 1; }
tail -3 gpurun_out/r3c_all.log
