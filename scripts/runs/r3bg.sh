#!/bin/bash
O=gpurun_out/r3bg; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py tests/test_gpu_rank.py tests/test_gpu_assembly.py tests/test_gpu_fullsize.py -q -m gpu -k "not item2vec" -p no:cacheprovider -x 2>&1 | tail -2
for R in 1 2 3; do
for W in "" "--workload youtube"; do
timeout 300 python bench.py $W --no-cpu-baseline --no-serving --no-roofline --phase predict --steps 200 --warmup 20 > $O/x.json 2> $O/x.err
python - <<P
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('$W', d.get('recommend_qps'))
P
done
done
timeout 120 ./goctr_amd/host/rank_bench --threads 1 --n 32,256,2048 --seconds 0.4 --coalesce 1 > $O/rank.json 2> $O/rank.err
python - <<P
import json
d=json.load(open('gpurun_out/r3bg/rank.json'))
print([(r['n'], r['threads'], round(r['rank_qps']), r['latency_us']['p50'], r['mismatched_calls']) for r in d['results']])
P
