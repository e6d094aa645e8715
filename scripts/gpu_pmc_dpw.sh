# HBM bytes of the wide-window sweep kernels on the star phylogeny with gaps: FETCH_SIZE and WRITE_SIZE in separate passes
set -x
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/pmc_dpw_fetch $O/pmc_dpw_write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_dpw" --output-format csv -d $O/pmc_dpw_fetch -o f -- python3 $GRAFT_REPO_ROOT/scripts/gpu_configs.py stargaps > $O/pmc_dpw_fetch.log 2>&1 || { tail -20 $O/pmc_dpw_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_dpw" --output-format csv -d $O/pmc_dpw_write -o w -- python3 $GRAFT_REPO_ROOT/scripts/gpu_configs.py stargaps > $O/pmc_dpw_write.log 2>&1 || { tail -20 $O/pmc_dpw_write.log; exit 1; }
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find $O/pmc_dpw_fetch -name "*counter_collection.csv") $(find $O/pmc_dpw_write -name "*counter_collection.csv") > $O/pmc_dpw_kernels.json
grep -A8 "k_dpw_chain\|k_dpw_blockM" $O/pmc_dpw_kernels.json | head -40
