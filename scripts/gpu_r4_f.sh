# round 4: the lean rank-order scan: parity, rank_no_lean 1, 0 in launch order, SQ counters
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 5 120 python scripts/gpu_small_repro.py || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msd or full_size or index_arrays or large_text or fixtures or random or column_shards" > gpurun_out/r4e_tests.log 2>&1 || { tail -40 gpurun_out/r4e_tests.log; exit 1; }
tail -3 gpurun_out/r4e_tests.log
bash scripts/gpu_trace_order.sh r4e_lean "k_msd_finish|k_rank_scan|k_tie_simple|k_runs|k_tie_groups|k_cand" gpu_c3_variants.py rank_no_lean 1,0 3 > gpurun_out/r4e_order_full.txt 2>&1
tail -12 gpurun_out/r4e_lean_order.txt
grep -o '"rank_no_lean": [0-9], "ms": [0-9.]*\|f_sum": [0-9]*' gpurun_out/trace_r4e_lean.log | paste - -
bash scripts/gpu_pmc_sq2.sh "k_rank_scan" gpu_c3_variants.py rank_no_lean 0 1 > gpurun_out/r4e_pmc.log 2>&1
tail -18 gpurun_out/pmc_sq2.txt
