#!/bin/bash
# round 6, session 39: gate and similarity weight of the attention forward as ONE factor (g (1 - g)) w where the chain launch's attention
# backward is their only reader (AttnArgs::fac, GOCTR_GATE_FAC): tests, then A/B against both arrays
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s39; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_multi.py tests/test_gpu_resume.py tests/test_gpu_comm.py tests/test_gpu_embtrain.py -q -m gpu -p no:cacheprovider -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
train() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], {k:v.get('avg_us') for k,v in d.get('kernels',{}).items() if isinstance(v,dict)})
P
}
for rep in 1 2 3 4; do
train fac1_$rep
train fac0_$rep GOCTR_GATE_FAC=0
done
train drv_fac1 
