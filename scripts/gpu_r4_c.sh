# round 4: pass 3 of the MSD sort persistent / fused with the scan's classification: parity, then msd_fuse 0, 1, 3 in launch order
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msd or full_size or index_arrays or large_text or fixtures or random" > gpurun_out/r4c_tests.log 2>&1 || { tail -40 gpurun_out/r4c_tests.log; exit 1; }
tail -3 gpurun_out/r4c_tests.log
bash scripts/gpu_trace_order.sh r4c_fuse "k_msd_finish|k_rank_scan|k_tie_simple|k_runs|k_tie_groups|k_cand|k_fuse|k_msd_defer" gpu_c3_variants.py msd_fuse 0,1 3
grep -o '"msd_fuse": [0-9], "ms": [0-9.]*\|f_sum": [0-9]*' gpurun_out/trace_r4c_fuse.log | paste - -
