# idle stretches of the GPU in one command: the gaps of more than MIN_MS between one kernel's end and the next one's start, with both names.
# usage: bash scripts/gpu_trace_gaps.sh TAG MIN_MS script.py [args ...]
TAG=$1; MIN=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/trace_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$TAG -o s -- python3 $GRAFT_REPO_ROOT/scripts/"$@" > $OUT/trace_$TAG.log 2>&1 || { tail -30 $OUT/trace_$TAG.log; exit 1; }
python3 - "$(find $OUT/trace_$TAG -name '*kernel_trace.csv' | head -1)" "$MIN" > $OUT/${TAG}_gaps.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for a, b in zip(rows, rows[1:]):
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e6
    if gap > float(sys.argv[2]):
        print(f'{(int(a["End_Timestamp"]) - t0) / 1e6:10.1f} ms  gap {gap:8.1f} ms  after {a["Kernel_Name"][:50]:50s} before {b["Kernel_Name"][:50]}')
PY
rm -rf $OUT/trace_$TAG
cat $OUT/${TAG}_gaps.txt
grep '"ms"' $OUT/trace_$TAG.log
