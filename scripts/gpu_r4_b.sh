# round 4: XCD placement in the pairs sort: parity tests that reach it, then C5 and the star phylogeny with gaps, placement off / on
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_group.py -m gpu -x -q -k "msd or pairs or part or sample or gapped or span or ignore" > gpurun_out/r4b_tests.log 2>&1 || { tail -30 gpurun_out/r4b_tests.log; exit 1; }
tail -3 gpurun_out/r4b_tests.log
for x in 0 3; do
  FBG_OPTS=msd_xcd=$x timeout -k 10 300 python scripts/gpu_configs.py c5 stargaps > gpurun_out/r4b_cfg_$x.log 2>&1 || { tail gpurun_out/r4b_cfg_$x.log; exit 1; }
  cat gpurun_out/r4b_cfg_$x.log
done
