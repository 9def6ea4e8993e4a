#!/bin/bash
O=gpurun_out/r3n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_embtrain.py -q -m gpu --timeout 600 -p no:cacheprovider --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 14 $O/pytest.log
for R in 1 0; do
GOCTR_NN_ROWS=$R timeout 300 python bench.py --workload youtube --train-emb 0.05 --no-cpu-baseline --no-serving > $O/yt_emb_rows$R.json 2> $O/yt_emb_rows$R.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3n/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
P
