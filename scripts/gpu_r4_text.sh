# round 4: k_write_text for rows with gaps through LDS: the tests that build such texts, then C5 and the star phylogeny with gaps
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_cli.py tests/test_group.py -m gpu -x -q -k "gap or ignore or span or fixtures or cli or random or c5 or C5 or record" > gpurun_out/r4t_tests.log 2>&1 || { tail -40 gpurun_out/r4t_tests.log; exit 1; }
tail -2 gpurun_out/r4t_tests.log
bash scripts/gpu_prof_cfg.sh c5
grep -E "k_write_text|k_gw_build|k_row_count" $(find gpurun_out/prof_cfg -name "*kernel_stats.csv" | head -1) | cut -c1-140
python scripts/gpu_stargaps.py 3 0 | cut -c1-400
