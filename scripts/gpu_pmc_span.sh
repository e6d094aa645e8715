# SQ counters of the odd-group kernel of the span scan (star phylogeny with gaps).  usage: bash scripts/gpu_pmc_span.sh [opt]
OPT=${1:-0}
set -x
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_IFETCH"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_span_$tag
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex "k_sp_odd_pairs" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_span_$tag -o s -- python3 $GRAFT_REPO_ROOT/scripts/gpu_stargaps.py 1 $OPT > $GRAFT_REPO_ROOT/gpurun_out/pmc_span.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_span.log; }
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/pmc_span_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: f"{v:.3e}" for k, v in acc.items()})
PY
done
