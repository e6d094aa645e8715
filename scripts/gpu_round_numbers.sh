# everything DESIGN.md / BASELINE.md quote: headline bench + rocprof stats, PMC traffic of the scan kernel,
# the other configs, the per-rank cost of the partitioned index
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/gpu_bench_full.sh || exit 1
bash scripts/gpu_pmc.sh || exit 1
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python scripts/gpu_configs.py c2 c5 c5gapped star > gpurun_out/configs.log 2>&1 || { tail gpurun_out/configs.log; exit 1; }
grep "{" gpurun_out/configs.log | cut -c1-500
timeout -k 10 500 python scripts/gpu_part_sim.py 2 > gpurun_out/part_sim2.log 2>&1 || { tail gpurun_out/part_sim2.log; exit 1; }
tail -1 gpurun_out/part_sim2.log
timeout -k 10 500 python scripts/gpu_part_sim.py 4 > gpurun_out/part_sim4.log 2>&1 || { tail gpurun_out/part_sim4.log; exit 1; }
tail -1 gpurun_out/part_sim4.log
timeout -k 10 500 python scripts/gpu_part_sim.py 8 1000000 --sequential > gpurun_out/part_sim8.log 2>&1 || { tail gpurun_out/part_sim8.log; exit 1; }
tail -1 gpurun_out/part_sim8.log
