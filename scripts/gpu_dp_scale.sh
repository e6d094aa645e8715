cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_dp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dp -o dp -- python3 $GRAFT_REPO_ROOT/scripts/debug/dp_scale.py 2>&1 | grep -E "^[0-9]+ " 
python3 - <<'PY'
import csv, os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_dp/dp_kernel_stats.csv")))
for r in rows[:10]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>4s} avg_ms={float(r['AverageNs'])/1e6:8.3f} max_ms={float(r['MaxNs'])/1e6:8.3f} tot={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
