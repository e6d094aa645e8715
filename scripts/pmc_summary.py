"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes as MI355X_MICROARCH.md
prescribes): counter unit KiB; on gfx950 FETCH_SIZE reports half the bytes of WIDE COALESCED STREAMING reads, so the read
figure is given raw and doubled (the guide's correction) -- and the doubled one enters hbm_bytes_per_launch only for
kernels that stream; for kernels that gather (8-byte reads at scattered places: 64-byte sectors counted as they are)
the raw figure does."""
import collections
import csv
import json
import sys

# kernels whose reads are scattered 8- / 16-byte accesses, not streams: no doubling (round-3 verdict: 17.7 GB "read" by
# k_tie_simple in 2.6 ms would have been 6.7 TB/s of gathers)
GATHER = ("k_tie_pairs", "k_tie_simple", "k_tie_groups", "k_tie_big", "k_runs", "k_sp_odd", "k_sp_chain", "k_sp_values", "k_sp_irr", "k_grs_scan_seg", "k_grs_long", "k_scan_exceptions",
          "k_bt_", "k_dp_bt", "k_block_group", "k_block_edges")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
c5 = "--c5" in sys.argv
star = "--star" in sys.argv
for path in args:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[name][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {"command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras  (scripts/gpu_r4_prof.sh)",
       "workload": {"rows": 1000, "cols": 1000000},
       **({"command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 scripts/gpu_configs.py c5  (scripts/gpu_round2_extra.sh)",
           "workload": {"rows": 256, "cols": 2000000, "gaps": "5 % in runs of 16", "N": "0.1 %", "ignore": "N", "builds_per_run": 2}} if c5 else {}),
       **({"command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 scripts/gpu_stargaps.py 1 0  (scripts/gpu_r4_prof.sh)",
           "workload": {"rows": 1000, "cols": 200000, "rows_are": "a star phylogeny, 1 % substitutions", "gaps": "2 % of the cells in runs of 8", "builds_per_run": 1}} if star else {}),
       "correction": "counter unit KiB; FETCH_SIZE doubled for gfx950 (MI355X_MICROARCH.md, HBM section) for the kernels that stream, raw for those that gather (read_correction)", "kernels": {}}
for name, d in sorted(acc.items()):
    k = {}
    if "FETCH_SIZE" in d:
        n = len(launches[name]["FETCH_SIZE"])
        k["launches"] = n
        k["FETCH_SIZE_KiB_per_launch"] = d["FETCH_SIZE"] / n
        k["hbm_read_bytes_raw"] = d["FETCH_SIZE"] / n * 1024
        k["hbm_read_bytes_x2"] = 2 * d["FETCH_SIZE"] / n * 1024
    if "WRITE_SIZE" in d:
        n = len(launches[name]["WRITE_SIZE"])
        k["WRITE_SIZE_KiB_per_launch"] = d["WRITE_SIZE"] / n
        k["hbm_write_bytes"] = d["WRITE_SIZE"] / n * 1024
    if "hbm_read_bytes_x2" in k and "hbm_write_bytes" in k:
        gather = name.startswith(GATHER)
        k["read_correction"] = "none (scattered reads)" if gather else "x2 (wide coalesced stream)"
        k["hbm_bytes_per_launch"] = (k["hbm_read_bytes_raw"] if gather else k["hbm_read_bytes_x2"]) + k["hbm_write_bytes"]
    out["kernels"][name] = k
print(json.dumps(out, indent=1))
