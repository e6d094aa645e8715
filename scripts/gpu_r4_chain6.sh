# round 4, second session: k_dp_chain6 (the walk over the byte matrices on six waves): parity, fuzz, bench line
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -m gpu -x -q -k "dp or nonelastic or repeatfree or segment or fixtures or cli" > gpurun_out/r4i_tests.log 2>&1 || { tail -40 gpurun_out/r4i_tests.log; exit 1; }
tail -2 gpurun_out/r4i_tests.log
timeout -k 10 200 python scripts/gpu_fuzz_dp.py 60 615000 > gpurun_out/r4i_fuzz_dp.log 2>&1 || { tail -20 gpurun_out/r4i_fuzz_dp.log; exit 1; }
tail -1 gpurun_out/r4i_fuzz_dp.log
timeout -k 10 400 python bench.py > gpurun_out/r4i_bench.json 2> gpurun_out/r4i_bench.err || { tail -20 gpurun_out/r4i_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4i_bench.json").read().strip().splitlines()[-1])
print("C3 ms/step", round(d["ms_per_step"], 2), "sweep", d["sweep"], "latency", d.get("latency_ms"))
for w in d["other_workloads"]:
    print(w["workload"][:60], round(w["ms_per_step"], 2), {k: v for k, v in w["stages_ms"].items() if v > 0.3}, "dp_kind", w.get("dp_kind"))
PY
