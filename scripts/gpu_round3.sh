# Round-3 measurement set (everything DESIGN.md / BASELINE.md quote).  usage: bash scripts/gpu_round3.sh [TAG]
# -> gpurun_out/<TAG>_bench.json, <TAG>_kernel_stats.csv, <TAG>_pmc_kernels.json, <TAG>_stargaps_kernel_stats.csv,
#    <TAG>_stargaps_pmc_kernels.json, <TAG>_gloo2.json
TAG=${1:-r03}
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -30 gpurun_out/${TAG}_bench.err; exit 1; }
cut -c1-600 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_$TAG.log 2>&1 || { tail -30 $R/gpurun_out/prof_$TAG.log; exit 1; }
cp $(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
find $R/gpurun_out/prof_$TAG -name "*kernel_trace*" -size +20M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${TAG}_$c
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-include-regex "k_rank_scan|k_msd_|k_tie_simple|k_copy_rows|k_row_count" --output-format csv -d $R/gpurun_out/pmc_${TAG}_$c -o s -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$TAG.log; exit 1; }
  cp $(find $R/gpurun_out/pmc_${TAG}_$c -name "*counter_collection.csv" | head -1) $R/gpurun_out/${TAG}_pmc_${c}_counter_collection.csv
done
python3 $R/scripts/pmc_summary.py $R/gpurun_out/${TAG}_pmc_FETCH_SIZE_counter_collection.csv $R/gpurun_out/${TAG}_pmc_WRITE_SIZE_counter_collection.csv > $R/gpurun_out/${TAG}_pmc_kernels.json
# similar rows with gaps (span_scan.hip): kernel stats and the FETCH / WRITE passes of its kernels
rm -rf $R/gpurun_out/prof_${TAG}_star
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_star -o s -- python3 $R/scripts/gpu_stargaps.py 2 0 > $R/gpurun_out/prof_${TAG}_star.log 2>&1 || { tail -30 $R/gpurun_out/prof_${TAG}_star.log; exit 1; }
cp $(find $R/gpurun_out/prof_${TAG}_star -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_stargaps_kernel_stats.csv
find $R/gpurun_out/prof_${TAG}_star -name "*kernel_trace*" -size +20M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${TAG}_star_$c
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-include-regex "k_sp_|k_pp_" --output-format csv -d $R/gpurun_out/pmc_${TAG}_star_$c -o s -- python3 $R/scripts/gpu_stargaps.py 1 0 > $R/gpurun_out/pmc_${TAG}_star.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}_star.log; exit 1; }
  cp $(find $R/gpurun_out/pmc_${TAG}_star_$c -name "*counter_collection.csv" | head -1) $R/gpurun_out/${TAG}_star_pmc_${c}_counter_collection.csv
done
python3 $R/scripts/pmc_summary.py --star $R/gpurun_out/${TAG}_star_pmc_FETCH_SIZE_counter_collection.csv $R/gpurun_out/${TAG}_star_pmc_WRITE_SIZE_counter_collection.csv > $R/gpurun_out/${TAG}_stargaps_pmc_kernels.json
cd $R
# two ranks sharing this GPU over gloo: the multi-process code path of bench.py (partitioned index, collectives, failure agreement)
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --cols-per-gpu 500000 > gpurun_out/${TAG}_gloo2.json 2> gpurun_out/${TAG}_gloo2.err || { tail -20 gpurun_out/${TAG}_gloo2.err; exit 1; }
cut -c1-500 gpurun_out/${TAG}_gloo2.json
