cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cfg=$1; shift
for e in "$@"; do
echo "== $e"
env $e timeout -k 10 400 python scripts/gpu_configs.py $cfg 2>&1 | tail -2 | cut -c1-600
done
