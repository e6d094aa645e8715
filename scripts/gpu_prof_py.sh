# rocprofv3 --kernel-trace --stats of a python script: gpu_prof_py.sh TAG script args...  -> gpurun_out/TAG_kernel_stats.csv
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o p -- python3 $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log; exit 1; }
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv
grep -v "^W\|amdgpu.ids" $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log | cut -c1-400
head -${LINES_SHOWN:-22} $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
