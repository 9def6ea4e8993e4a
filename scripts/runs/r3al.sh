#!/bin/bash
O=gpurun_out/r3al; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py tests/test_gpu_fullsize.py tests/test_gpu_rank.py -q -m gpu -k "not item2vec" -p no:cacheprovider -x 2>&1 | tail -3
for R in 1 2 3; do
for P in 1 0; do
GOCTR_FWD_PERSIST=$P timeout 300 python bench.py --no-cpu-baseline --no-serving --no-roofline --phase predict --steps 200 --warmup 20 > $O/p$P.json 2> $O/p$P.err
python - <<PY
import json
d=json.loads(open('$O/p$P.json').read().strip().splitlines()[-1]); print('persist=$P', d.get('recommend_qps'), d.get('recommend_qps_keys'))
PY
done
done
for P in 1 0; do
GOCTR_FWD_PERSIST=$P timeout 300 python bench.py --workload youtube --no-cpu-baseline --no-serving --no-roofline --phase predict --steps 200 --warmup 20 > $O/y$P.json 2> $O/y$P.err
python - <<PY
import json
d=json.loads(open('$O/y$P.json').read().strip().splitlines()[-1]); print('youtube persist=$P', d.get('recommend_qps'), d.get('recommend_qps_keys'))
PY
done
