set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/parity1.log
cat gpurun_out/parity1.log
