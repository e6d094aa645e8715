# round 4: the wide-window sweep with and without the wide matrices over window sizes 1024 .. 16384 (made-up f), and on the star
# phylogeny with gaps; the sweep tests first
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dp_sweep" > gpurun_out/r04_dpw_tests.log 2>&1 || { tail -40 gpurun_out/r04_dpw_tests.log; exit 1; }
tail -2 gpurun_out/r04_dpw_tests.log
bash scripts/gpu_trace_order.sh r04_dpw_windows "k_dpw" gpu_dpw_windows.py
grep "max_ext" gpurun_out/trace_r04_dpw_windows.log > gpurun_out/r04_dpw_windows.txt
bash scripts/gpu_trace_order.sh r04_dpw_stargaps "k_dpw" gpu_dpw_probe.py 1 0 1 0
grep -E "extensions|dpw_matrix" gpurun_out/trace_r04_dpw_stargaps.log
