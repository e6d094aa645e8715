set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
bash scripts/gpu_trace_order.sh r4g_probe "k_dpw" gpu_dpw_probe.py 0 62 126 190 318 446 510
grep -E "extensions|dpw_matrix" gpurun_out/trace_r4g_probe.log
