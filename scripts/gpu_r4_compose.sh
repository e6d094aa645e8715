# round 4: k_dp_compose with a 4 x 8 register tile: parity, fuzz, C5 kernel times
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -m gpu -x -q -k "dp or nonelastic or repeatfree or segment or fixtures or cli" > gpurun_out/r4k_tests.log 2>&1 || { tail -40 gpurun_out/r4k_tests.log; exit 1; }
tail -2 gpurun_out/r4k_tests.log
timeout -k 10 200 python scripts/gpu_fuzz_dp.py 60 715000 > gpurun_out/r4k_fuzz_dp.log 2>&1 || { tail -20 gpurun_out/r4k_fuzz_dp.log; exit 1; }
tail -1 gpurun_out/r4k_fuzz_dp.log
bash scripts/gpu_prof_cfg.sh c5
grep -E "k_dp_|k_bt_lift|k_dp_keys" $(find gpurun_out/prof_cfg -name "*kernel_stats.csv" | head -1) | cut -c1-120
