set -x
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"; do
n=$(echo $c | cut -c1-6)
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_bp_$n
timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex "k_bp_" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_bp_$n -o s -- python3 $GRAFT_REPO_ROOT/scripts/gpu_configs.py c5 > $GRAFT_REPO_ROOT/gpurun_out/pmc_bp.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_bp.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, os, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_bp_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:14]][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[r["Kernel_Name"][:14]][r["Counter_Name"]] += 1
for k, d in acc.items():
    print(k, {c: f"{v/ max(1,cnt[k][c]):.4g}" for c, v in d.items()})
PY
