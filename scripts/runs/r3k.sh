#!/bin/bash
# round 3, GPU call K: slot kernel with 4 components per lane; merged state-preparation launch
O=gpurun_out/r3k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_embtrain.py tests/test_gpu_comm.py tests/test_gpu_ctr.py tests/test_gpu_pipeline.py tests/test_gpu_resume.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
for V in 4 1; do
GOCTR_EMB_SLOT_VEC=$V timeout 300 python bench.py --train-emb 0.05 --no-cpu-baseline --no-serving > $O/din_emb_v$V.json 2> $O/din_emb_v$V.err
GOCTR_EMB_SLOT_VEC=$V timeout 300 python bench.py --workload youtube --train-emb 0.05 --no-cpu-baseline --no-serving > $O/yt_emb_v$V.json 2> $O/yt_emb_v$V.err
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serving > $O/b20.json 2> $O/b20.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3k/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
P
