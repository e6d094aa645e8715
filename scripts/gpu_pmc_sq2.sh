# SQ counters of the kernels matching $1 in one command (script + args after it), two passes.  usage: bash scripts/gpu_pmc_sq2.sh REGEX script.py [args]
set -x
RE=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for pass in a b; do
  if [ $pass = a ]; then C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; else C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; fi
  rm -rf $OUT/pmc_sq_$pass
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-include-regex "$RE" --output-format csv -d $OUT/pmc_sq_$pass -o s -- python3 $GRAFT_REPO_ROOT/scripts/"$@" > $OUT/pmc_sq_$pass.log 2>&1 || { tail -20 $OUT/pmc_sq_$pass.log; exit 1; }
done
python3 - <<'PY' > $OUT/pmc_sq2.txt
import csv, glob, os, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for p in "ab":
    f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + f"/gpurun_out/pmc_sq_{p}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Kernel_Name"][:40]][r["Counter_Name"]] += 1
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f"    {c:24s} {v / max(1, n[k][c]):14.5g}   (per dispatch, {n[k][c]} dispatches)")
PY
cat $OUT/pmc_sq2.txt
