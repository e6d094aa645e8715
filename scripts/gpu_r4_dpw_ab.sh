# round 4 (second session): the wide-window sweep without wide matrices (k_dpw_blockY / k_dpw_chain2): parity, fuzz, then the
# star phylogeny with gaps with dpw_matrix = 1, 0 in launch order
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dp_sweep" > gpurun_out/r4f_tests.log 2>&1 || { tail -40 gpurun_out/r4f_tests.log; exit 1; }
tail -3 gpurun_out/r4f_tests.log
timeout -k 10 200 python scripts/gpu_fuzz_dp.py 90 515000 > gpurun_out/r4f_fuzz_dp.log 2>&1 || { tail -20 gpurun_out/r4f_fuzz_dp.log; exit 1; }
tail -2 gpurun_out/r4f_fuzz_dp.log
bash scripts/gpu_trace_order.sh r4f_m1 "k_dpw|k_dp_keys|k_bt" gpu_stargaps.py 2 0 dpw_matrix=1
bash scripts/gpu_trace_order.sh r4f_m0 "k_dpw|k_dp_keys|k_bt" gpu_stargaps.py 2 0 dpw_matrix=0
