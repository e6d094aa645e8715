# round 4, first call: MSD tests with the XCD placement, then the four placements timed (kernel trace in launch order)
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msd or index_arrays or large_text" > gpurun_out/r4a_tests.log 2>&1 || { tail -30 gpurun_out/r4a_tests.log; exit 1; }
tail -3 gpurun_out/r4a_tests.log
bash scripts/gpu_trace_order.sh r4a_xcd "k_msd|k_rank_scan|k_tie_simple" gpu_c3_variants.py msd_xcd 0,1,2,3 3
