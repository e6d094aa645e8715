# round 4: the kernels around the text of rows with gaps (k_write_text through LDS, k_ss_sample with 8-byte loads, k_gw_build four
# windows per wave): the tests that build such texts, then C5 under the kernel trace
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_cli.py tests/test_group.py -m gpu -x -q -k "gap or ignore or span or fixtures or cli or random or c5 or C5 or record or partition" > gpurun_out/r4t_tests.log 2>&1 || { tail -40 gpurun_out/r4t_tests.log; exit 1; }
tail -2 gpurun_out/r4t_tests.log
timeout -k 10 300 python scripts/gpu_fuzz.py 120 1600000 > gpurun_out/r4t_fuzz.log 2>&1 || { tail -20 gpurun_out/r4t_fuzz.log; exit 1; }
tail -1 gpurun_out/r4t_fuzz.log
bash scripts/gpu_prof_cfg.sh c5
grep -E "k_write_text|k_gw_build|k_ss_sample" $(find gpurun_out/prof_cfg -name "*kernel_stats.csv" | head -1) | cut -c1-60,150-260
