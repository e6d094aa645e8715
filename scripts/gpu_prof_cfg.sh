set -x
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_cfg
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg -o cfg -- python3 $GRAFT_REPO_ROOT/scripts/gpu_configs.py $1 > $GRAFT_REPO_ROOT/gpurun_out/prof_cfg.log 2>&1 || { tail -30 $GRAFT_REPO_ROOT/gpurun_out/prof_cfg.log; exit 1; }
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_cfg.log | cut -c1-300
find $GRAFT_REPO_ROOT/gpurun_out/prof_cfg -name "*kernel_trace*" -size +20M -delete
