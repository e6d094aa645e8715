# round 4 profiles: kernel-trace stats of bench.py (C3), FETCH_SIZE / WRITE_SIZE passes over every kernel of the step; the same for
# BASELINE config 5 and the star phylogeny with gaps.  Summaries -> gpurun_out/r04_*
set -x
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/p4
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p4/c3 -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/r04_prof_bench.log 2>&1 || { tail -20 $OUT/r04_prof_bench.log; exit 1; }
cp $(find $OUT/p4/c3 -name '*kernel_stats.csv' | head -1) $OUT/r04_kernel_stats.csv
grep "^{\"metric\"" $OUT/r04_prof_bench.log > $OUT/r04_prof_bench.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $OUT/p4/c3_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/r04_pmc_$c.log 2>&1 || { tail -20 $OUT/r04_pmc_$c.log; exit 1; }
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find $OUT/p4/c3_FETCH_SIZE -name '*counter_collection.csv') $(find $OUT/p4/c3_WRITE_SIZE -name '*counter_collection.csv') > $OUT/r04_pmc_kernels.json
# config 5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p4/c5 -o c5 -- python3 $GRAFT_REPO_ROOT/scripts/gpu_configs.py c5 > $OUT/r04_c5.log 2>&1 || { tail -20 $OUT/r04_c5.log; exit 1; }
cp $(find $OUT/p4/c5 -name '*kernel_stats.csv' | head -1) $OUT/r04_c5_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $OUT/p4/c5_$c -o p -- python3 $GRAFT_REPO_ROOT/scripts/gpu_configs.py c5 > $OUT/r04_c5_pmc_$c.log 2>&1 || { tail -20 $OUT/r04_c5_pmc_$c.log; exit 1; }
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py --c5 $(find $OUT/p4/c5_FETCH_SIZE -name '*counter_collection.csv') $(find $OUT/p4/c5_WRITE_SIZE -name '*counter_collection.csv') > $OUT/r04_c5_pmc_kernels.json
# star phylogeny with gaps
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p4/sg -o sg -- python3 $GRAFT_REPO_ROOT/scripts/gpu_stargaps.py 3 0 > $OUT/r04_stargaps.log 2>&1 || { tail -20 $OUT/r04_stargaps.log; exit 1; }
cp $(find $OUT/p4/sg -name '*kernel_stats.csv' | head -1) $OUT/r04_stargaps_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $OUT/p4/sg_$c -o p -- python3 $GRAFT_REPO_ROOT/scripts/gpu_stargaps.py 1 0 > $OUT/r04_sg_pmc_$c.log 2>&1 || { tail -20 $OUT/r04_sg_pmc_$c.log; exit 1; }
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py --star $(find $OUT/p4/sg_FETCH_SIZE -name '*counter_collection.csv') $(find $OUT/p4/sg_WRITE_SIZE -name '*counter_collection.csv') > $OUT/r04_stargaps_pmc_kernels.json
# the same rows, ten times the columns: 2 * 10^9 cells (the span scan's slots with the flags in the key word)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p4/sb -o sb -- python3 $GRAFT_REPO_ROOT/scripts/gpu_span_big.py 2 > $OUT/r04_span_2e9.log 2>&1 || { tail -20 $OUT/r04_span_2e9.log; exit 1; }
cp $(find $OUT/p4/sb -name '*kernel_stats.csv' | head -1) $OUT/r04_span_2e9_kernel_stats.csv
find $OUT/p4 -name "*kernel_trace*" -delete
rm -rf $OUT/p4/sg_*
rm -rf $OUT/p4/c3_* $OUT/p4/c5_*
grep "^{" $OUT/r04_c5.log | cut -c1-400; grep '"ms"' $OUT/r04_stargaps.log | cut -c1-300; grep '"ms"' $OUT/r04_span_2e9.log | cut -c1-300
head -12 $OUT/r04_kernel_stats.csv | cut -c1-120
