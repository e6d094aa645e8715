set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -30 gpurun_out/smoke.log; exit 1; }
tail -3 gpurun_out/smoke.log
timeout -k 10 300 python bench.py --rows 1000 --cols-per-gpu 20000 --steps 2 --warmup 1 --cpu-sample-cols 5000 > gpurun_out/bench_small.log 2>&1 || { tail -30 gpurun_out/bench_small.log; exit 1; }
tail -3 gpurun_out/bench_small.log
timeout -k 10 600 python bench.py --rows 1000 --cols-per-gpu 200000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_200k.log 2>&1 || { tail -30 gpurun_out/bench_200k.log; exit 1; }
tail -3 gpurun_out/bench_200k.log
