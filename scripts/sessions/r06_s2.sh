#!/bin/bash
# round 6, session 2: XCD-affine row dealing (chain tiles, attention groups) A/B + the CTR tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s2; mkdir -p $O
for rep in 1 2; do for x in 0 1; do
GOCTR_XCD_AFFINE=$x timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/din_x${x}_$rep.json 2> $O/din_x${x}_$rep.err
python - <<P
import json
d=json.loads(open('$O/din_x${x}_$rep.json').read().strip().splitlines()[-1]); print('affine=$x rep $rep', d['value'], d['ms_per_step'], 'qps', d.get('recommend_qps'), {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
done; done
for x in 0 1; do echo "affine=$x"; GOCTR_XCD_AFFINE=$x timeout 120 python scripts/dbg_chain.py 2>&1 | grep -h "phases\|tn_multi\|tn " | tail -4; done
GOCTR_XCD_AFFINE=1 timeout 300 python bench.py --workload youtube --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/yt_x1.json 2> $O/yt_x1.err
GOCTR_XCD_AFFINE=0 timeout 300 python bench.py --workload youtube --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/yt_x0.json 2> $O/yt_x0.err
python - <<P
import json
for x in (0,1):
    d=json.loads(open('$O/yt_x%d.json'%x).read().strip().splitlines()[-1]); print('youtube affine=%d'%x, d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
timeout 1200 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -5
