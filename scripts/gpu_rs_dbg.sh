cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in 0 1 2 3 4 7; do
FBG_RS_DBG=$d timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/dbg.log 2>&1 || { tail -5 gpurun_out/dbg.log; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/dbg.log").read().strip().splitlines()[-1])
print("dbg=$d", "rank_kernel", round(d["stages_ms_per_step"]["rank_kernel"], 2), "rank_scan", round(d["stages_ms_per_step"]["rank_scan"], 2))
PY
done
