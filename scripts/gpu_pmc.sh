set -x
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch $GRAFT_REPO_ROOT/gpurun_out/pmc_write
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_rank_scan|k_scan_stream|k_apply_groups0" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_rank_scan|k_scan_stream|k_apply_groups0" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log; exit 1; }
ls -R $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch | head; head -5 $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch/*counter_collection.csv
