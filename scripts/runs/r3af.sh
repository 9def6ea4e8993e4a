#!/bin/bash
O=gpurun_out/r3af; mkdir -p $O
for R in 1 2; do
timeout 300 python bench.py --workload item2vec --no-cpu-baseline > $O/i2v.json 2> $O/i2v.err
python - <<P
import json
d=json.loads(open('$O/i2v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
P
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_w2v.py -q -m gpu -k "w2v or item2vec or hogwild" -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -8
