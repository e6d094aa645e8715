# kernel dispatches of one command in launch order (name, microseconds), for the probe variants of a kernel.
# usage: bash scripts/gpu_trace_order.sh TAG PATTERN script.py [args ...]
TAG=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/trace_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$TAG -o s -- python3 $GRAFT_REPO_ROOT/scripts/"$@" > $OUT/trace_$TAG.log 2>&1 || { tail -30 $OUT/trace_$TAG.log; exit 1; }
python3 - "$(find $OUT/trace_$TAG -name '*kernel_trace.csv' | head -1)" "$PAT" > $OUT/${TAG}_order.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    if re.search(sys.argv[2], r["Kernel_Name"]):
        print(f'{r["Kernel_Name"][:60]:60s} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:10.1f} us')
PY
rm -rf $OUT/trace_$TAG
cat $OUT/${TAG}_order.txt
tail -3 $OUT/trace_$TAG.log
