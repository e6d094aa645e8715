cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for e in "$@"; do
env $e timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 > gpurun_out/cmp.log 2>&1 || { tail -5 gpurun_out/cmp.log; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/cmp.log").read().strip().splitlines()[-1])
print("$e", round(d["ms_per_step"], 2), d["config"]["blocks"], {k: round(v, 2) for k, v in d["stages_ms_per_step"].items()})
PY
done
