set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<'PY'
import numpy as np, time, sys
sys.path.insert(0,'.')
from bench import synth_msa_host
import os
m,n=1000,int(os.environ.get('E2E_COLS','200000'))
a=synth_msa_host(m,n,n)
t=time.time()
with open('/tmp/c3s.fasta','wb') as fh:
    for i in range(m):
        fh.write(b'>r%d\n'%i); fh.write(a[i].tobytes()); fh.write(b'\n')
print('fasta written', time.time()-t, a.shape)
PY
S=$(date +%s.%N); FBG_TIMING=1 ./founderblockgraphs_amd/founderblockgraph --input /tmp/c3s.fasta --output /tmp/c3s.xgfa --elastic --gfa -p 2> gpurun_out/cli_e2e.log; E=$(date +%s.%N); python3 -c "print('wall seconds', $E - $S)"; tail -25 gpurun_out/cli_e2e.log | grep -E "Time taken|timing|Input MSA|optimal"; ls -la /tmp/c3s.xgfa; head -c 300 /tmp/c3s.xgfa | head -3 | cut -c1-100
