set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "span_scan" > gpurun_out/r3b_span.log 2>&1 || { tail -60 gpurun_out/r3b_span.log; exit 1; }
tail -3 gpurun_out/r3b_span.log
timeout -k 10 300 python scripts/gpu_stargaps.py 3 0 > gpurun_out/r3b_star.log 2>&1 || { tail -30 gpurun_out/r3b_star.log; exit 1; }
cat gpurun_out/r3b_star.log
timeout -k 10 300 python scripts/gpu_stargaps.py 2 -1 > gpurun_out/r3b_star_rec.log 2>&1 || { tail -30 gpurun_out/r3b_star_rec.log; exit 1; }
cat gpurun_out/r3b_star_rec.log
