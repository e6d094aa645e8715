# SQ counters of the kernels matching a regex while scripts/gpu_configs.py <config> runs:
# usage: gpu_pmc_kernel.sh REGEX config   -> prints per-kernel sums
set -x
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_k
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-include-regex "$1" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_k -o s -- python3 $GRAFT_REPO_ROOT/scripts/gpu_configs.py $2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_k.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/pmc_k.log; exit 1; }
python3 - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_k/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:40]].add(r["Dispatch_Id"])
for k, d in acc.items():
    print(k, "launches", len(n[k]), {c: f"{v / len(n[k]):.4g}" for c, v in d.items()})
PY
