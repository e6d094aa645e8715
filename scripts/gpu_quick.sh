set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/parity.log || { cat gpurun_out/parity.log; exit 1; }
tail -3 gpurun_out/parity.log
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1 || { tail -30 gpurun_out/bench_quick.log; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_quick.log").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["blocks"], d["stages_ms_per_step"])
PY
