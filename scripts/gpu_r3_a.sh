# round 3, first call: the new tests (ADVICE fixes, full-size independent paths), then the whole GPU suite
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "independent_paths or fresh_context" > gpurun_out/r3a_new.log 2>&1 || { tail -40 gpurun_out/r3a_new.log; exit 1; }
tail -5 gpurun_out/r3a_new.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_all.log 2>&1 || { tail -40 gpurun_out/r3a_all.log; exit 1; }
tail -5 gpurun_out/r3a_all.log
