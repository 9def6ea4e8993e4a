#!/bin/bash
O=gpurun_out/r3r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_embtrain.py tests/test_gpu_comm.py tests/test_gpu_ctr.py tests/test_gpu_fullsize.py::test_cfg3_din_full_size_step_vs_oracle tests/test_gpu_pipeline.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
for C in 1 0; do
GOCTR_EMB_DPV_CHAIN=$C timeout 300 python bench.py --train-emb 0.05 --no-cpu-baseline --no-serving > $O/din_emb_c$C.json 2> $O/din_emb_c$C.err
done
timeout 300 python bench.py --no-cpu-baseline --no-serving > $O/din.json 2> $O/din.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3r/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
P
