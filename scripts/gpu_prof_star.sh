# kernel stats of the star-phylogeny-with-gaps workload.  usage: bash scripts/gpu_prof_star.sh TAG [span_scan option]
TAG=${1:-star}
OPT=${2:-0}
set -x
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o s -- python3 $GRAFT_REPO_ROOT/scripts/gpu_stargaps.py 2 $OPT > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1 || { tail -30 $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log; exit 1; }
cp $(find $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv
find $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -name "*kernel_trace*" -size +20M -delete
head -25 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log
