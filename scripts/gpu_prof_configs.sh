# per-kernel times of the workloads beside the headline: rocprofv3 --kernel-trace --stats of scripts/gpu_configs.py <config>
# usage: gpu_prof_configs.sh TAG config...   -> gpurun_out/TAG_<config>_kernel_stats.csv
set -x
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$c -o p -- python3 $GRAFT_REPO_ROOT/scripts/gpu_configs.py $c > $GRAFT_REPO_ROOT/gpurun_out/prof_$c.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/prof_$c.log; exit 1; }
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$c -name "*kernel_stats.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_${c}_kernel_stats.csv
  grep "{" $GRAFT_REPO_ROOT/gpurun_out/prof_$c.log | cut -c1-600
  head -25 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_${c}_kernel_stats.csv | cut -c1-160
done
