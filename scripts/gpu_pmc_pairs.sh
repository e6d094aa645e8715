cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-30)
  rm -rf $R/gpurun_out/pmc1_$tag
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex "k_sp_odd_pairs<28" --output-format csv -d $R/gpurun_out/pmc1_$tag -o s -- python3 $R/scripts/gpu_stargaps.py 1 0 > $R/gpurun_out/pmc1.log 2>&1 || tail -3 $R/gpurun_out/pmc1.log
  f=$(find $R/gpurun_out/pmc1_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: f"{v:.3e}" for k, v in acc.items()})
PY
done
