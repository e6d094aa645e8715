# round 4: pass 3 with 2048 bins and binned-order counting: parity (incl. similar rows: crowded bins), times
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 5 120 python scripts/gpu_small_repro.py || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msd or full_size or index_arrays or large_text or fixtures or random or column_shards or similar or pure or star" > gpurun_out/r4g_tests.log 2>&1 || { tail -40 gpurun_out/r4g_tests.log; exit 1; }
tail -3 gpurun_out/r4g_tests.log
bash scripts/gpu_trace_order.sh r4g "k_msd_finish\(|k_rank_scan|k_tie|k_runs|k_cand" gpu_c3_once.py 3 > gpurun_out/r4g_order_full.txt 2>&1
tail -9 gpurun_out/r4g_order.txt
grep -o '"ms": [0-9.]*\|f_sum": [0-9]*' gpurun_out/trace_r4g.log | paste - -
timeout -k 10 300 python scripts/gpu_configs.py star c2 > gpurun_out/r4g_cfg.log 2>&1 || { tail gpurun_out/r4g_cfg.log; exit 1; }
cat gpurun_out/r4g_cfg.log
