#!/bin/bash
O=gpurun_out/r3m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_embtrain.py tests/test_gpu_comm.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 8 $O/pytest.log
timeout 300 python bench.py --train-emb 0.05 --no-cpu-baseline --no-serving > $O/din_emb.json 2> $O/din_emb.err
timeout 300 python bench.py --workload youtube --train-emb 0.05 --no-cpu-baseline --no-serving > $O/yt_emb.json 2> $O/yt_emb.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3m/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
P
PASSES="kt" timeout 300 bash scripts/prof_workload.sh dinemb --train-emb 0.05 > /dev/null 2>&1
find gpurun_out/p_dinemb -name '*_kernel_stats.csv' | xargs -I{} sh -c 'cut -c1-110 {} | head -12'
