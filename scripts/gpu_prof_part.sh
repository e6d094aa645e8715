set -x
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_part
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_part -o p -- python3 $GRAFT_REPO_ROOT/scripts/gpu_part_sim.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_part.log 2>&1 || { tail -30 $GRAFT_REPO_ROOT/gpurun_out/prof_part.log; exit 1; }
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_part.log | cut -c1-300
find $GRAFT_REPO_ROOT/gpurun_out/prof_part -name "*kernel_trace*" -delete
