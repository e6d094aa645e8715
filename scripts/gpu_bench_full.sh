set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/bench_full.log 2>&1 || { tail -30 gpurun_out/bench_full.log; exit 1; }
tail -2 gpurun_out/bench_full.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1.log 2>&1 || { tail -30 $GRAFT_REPO_ROOT/gpurun_out/prof_r1.log; exit 1; }
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_r1.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -name "*stats*" | head; find $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -name "*kernel_trace*" -size +20M -delete
