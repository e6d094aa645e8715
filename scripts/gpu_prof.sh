set -x
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_cur
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cur -o cur -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_cur.log 2>&1 || { tail -30 $GRAFT_REPO_ROOT/gpurun_out/prof_cur.log; exit 1; }
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_cur.log | cut -c1-300
find $GRAFT_REPO_ROOT/gpurun_out/prof_cur -name "*kernel_trace*" -size +20M -delete
