# per-kernel times of the partition builds of C4 on one GPU (16 builds per scan: 8 partitions, two passes)
set -x
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_c4
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4 -o p -- python3 $GRAFT_REPO_ROOT/scripts/gpu_c4_one_gpu.py > $GRAFT_REPO_ROOT/gpurun_out/prof_c4.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/prof_c4.log; exit 1; }
cp $(find $GRAFT_REPO_ROOT/gpurun_out/prof_c4 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/${1:-r02}_c4_kernel_stats.csv
find $GRAFT_REPO_ROOT/gpurun_out/prof_c4 -name "*kernel_trace*" -size +20M -delete
grep "{" $GRAFT_REPO_ROOT/gpurun_out/prof_c4.log
head -16 $GRAFT_REPO_ROOT/gpurun_out/${1:-r02}_c4_kernel_stats.csv | cut -c1-150
