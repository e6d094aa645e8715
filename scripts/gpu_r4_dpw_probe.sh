set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dp_sweep" > gpurun_out/r4g_tests.log 2>&1 || { tail -40 gpurun_out/r4g_tests.log; exit 1; }
tail -2 gpurun_out/r4g_tests.log
bash scripts/gpu_trace_order.sh r4g_probe "k_dpw" gpu_dpw_probe.py 0 1 2 4
grep -E "extensions|dpw_matrix" gpurun_out/trace_r4g_probe.log
