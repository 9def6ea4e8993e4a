#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_embtrain.py tests/test_gpu_comm.py tests/test_gpu_rank.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
for k in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_ctr.py::test_concurrent_handles_from_several_threads tests/test_gpu_rank.py::test_serving_while_training_the_same_model -q -m gpu -p no:cacheprovider 2>&1 | tail -1; done
for V in auto 4 1; do
E2=""; [ $V != auto ] && E2="GOCTR_EMB_SLOT_VEC=$V"
env $E2 timeout 300 python bench.py --train-emb 0.05 --no-cpu-baseline --no-serving > $O/din_emb_v$V.json 2> $O/din_emb_v$V.err
env $E2 timeout 300 python bench.py --workload youtube --train-emb 0.05 --no-cpu-baseline --no-serving > $O/yt_emb_v$V.json 2> $O/yt_emb_v$V.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3l/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
P
