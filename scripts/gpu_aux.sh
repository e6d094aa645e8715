set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/parity.log
rc=$?
cat gpurun_out/parity.log
[ $rc -eq 0 ] || exit $rc
p() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), d['config']['blocks'], {k: round(v,2) for k,v in d['stages_ms_per_step'].items()})"; }
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | p
FBG_NO_AUX_STREAM=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | p
done
