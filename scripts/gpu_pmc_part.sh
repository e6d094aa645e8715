set -x
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_part
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-include-regex "${1:-k_pp_pack_split}" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_part -o s -- python3 $GRAFT_REPO_ROOT/scripts/gpu_part_sim.py 8 1000000 --sequential > $GRAFT_REPO_ROOT/gpurun_out/pmc_part.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/pmc_part.log; exit 1; }
python3 - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_part/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:40]][r["Counter_Name"]] += 1
for k, d in acc.items():
    print(k, {c: f"{v / n[k][c]:.4g}" for c, v in d.items()})
PY
