# round 4, second session: the whole GPU suite, smoke, the bench line
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4h_tests.log 2>&1 || { tail -40 gpurun_out/r4h_tests.log; exit 1; }
tail -3 gpurun_out/r4h_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python bench.py > gpurun_out/r4h_bench.json 2> gpurun_out/r4h_bench.err || { tail -20 gpurun_out/r4h_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4h_bench.json").read().strip().splitlines()[-1])
print("C3 ms/step", round(d["ms_per_step"], 2), "boundary", d.get("boundary_ms"), "latency", d.get("latency_ms"))
for w in d["other_workloads"]:
    print(w["workload"][:60], round(w["ms_per_step"], 2), {k: v for k, v in w["stages_ms"].items() if v > 0.3}, "dp_kind", w.get("dp_kind"))
PY
