cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python "$@" 2>&1 | tail -20
