#!/bin/bash
O=gpurun_out/r3bh; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rank.py tests/test_gpu_assembly.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -2
for R in 1 2 3; do
for V in "GOCTR_SERVE_PIPELINE=1" "GOCTR_SERVE_PIPELINE=0" "GOCTR_SERVE_CHUNK=32768" "GOCTR_SERVE_CHUNK=8192"; do
env $V timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 > $O/x.json 2> $O/x.err
python - <<P
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('$V', d.get('recommend_qps_keys'), d.get('recommend_qps'))
P
done
done
