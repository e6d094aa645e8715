set -x
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_msd_$c
timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex "k_msd_|k_tie_simple|k_row_count|k_copy_rows" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_msd_$c -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_msd.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_msd.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, os, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_msd_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:24]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:24]][r["Counter_Name"]] += 1
for k, d in acc.items():
    print(k, {c: f"{v / n[k][c] / 1e6:.3f} GB(KiB-units/1e6)" for c, v in d.items()})
PY
