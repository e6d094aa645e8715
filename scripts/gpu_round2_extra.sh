# Round-2 measurements beside the headline: per-kernel times of C5 and the star input, HBM traffic of C5's kernels,
# k_dp_chain with the sweep not overlapped.  usage: bash scripts/gpu_round2_extra.sh [TAG]
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
bash scripts/gpu_prof_configs.sh $TAG c5 star > gpurun_out/${TAG}_prof_configs.log 2>&1 || { tail -20 gpurun_out/${TAG}_prof_configs.log; exit 1; }
grep "^{" gpurun_out/${TAG}_prof_configs.log | cut -c1-500
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_c5_$c
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-include-regex "k_pp_|k_grs_|k_gw_build|k_write_text|k_dp_blockW|k_dp_compose" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_c5_$c -o s -- python3 $GRAFT_REPO_ROOT/scripts/gpu_configs.py c5 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_c5.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_c5.log; exit 1; }
  cp $(find $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_c5_$c -name "*counter_collection.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_c5_${c}_counter_collection.csv
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py --c5 gpurun_out/${TAG}_pmc_c5_FETCH_SIZE_counter_collection.csv gpurun_out/${TAG}_pmc_c5_WRITE_SIZE_counter_collection.csv > gpurun_out/${TAG}_pmc_c5_kernels.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_pmc_c5_kernels.json"))
for k, v in d["kernels"].items():
    print(k[:40], v.get("launches"), round(v.get("hbm_read_bytes_x2", 0) / 1e9, 2), "GB read (x2)", round(v.get("hbm_write_bytes", 0) / 1e9, 2), "GB written")
PY
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_serial
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_serial -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --serial-sweep > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_serial.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_serial.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_serial/**/*kernel_trace.csv", recursive=True)[0]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "k_dp_chain" in r["Kernel_Name"]]
print("k_dp_chain, sweep not overlapped (us):", [round(x) for x in d])
open("$GRAFT_REPO_ROOT/gpurun_out/${TAG}_dp_chain_serial.txt", "w").write("k_dp_chain durations with --serial-sweep (us): " + str([round(x) for x in d]) + "\n")
PY
