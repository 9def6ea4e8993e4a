#!/bin/bash
# round 6, session 43: att0's sum and Adam update inside the weight-gradient launch (ctr_chain_x3.h att0_early_body, GOCTR_ATT0_EARLY): the
# last launch's attention wavefronts need no flag.  Tests, then A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s43; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py tests/test_gpu_fullsize.py tests/test_gpu_resume.py tests/test_gpu_multi.py tests/test_gpu_comm.py -q -m gpu -p no:cacheprovider -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
train() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], {k:v.get('avg_us') for k,v in d.get('kernels',{}).items() if isinstance(v,dict)})
P
}
for rep in 1 2 3 4; do
train early_$rep
train flag_$rep GOCTR_ATT0_EARLY=0
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serving > $O/drv.json 2> $O/drv.err; python -c "
import json; d=json.loads(open('$O/drv.json').read().strip().splitlines()[-1]); print('driver flags', d['value'], d['ms_per_step'], d['timed_regions_ms'])"
