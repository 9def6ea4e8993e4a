#!/bin/bash
O=gpurun_out/r3am; mkdir -p $O
for R in 1 2; do
for G in 4 8 16; do
GOCTR_PRED_GROUP=$G timeout 300 python bench.py --no-cpu-baseline --no-serving --steps 200 --warmup 20 > $O/g$G.json 2> $O/g$G.err
python - <<PY
import json
d=json.loads(open('$O/g$G.json').read().strip().splitlines()[-1]); print('group=$G', d['value'], d.get('recommend_qps'), d.get('recommend_qps_keys'))
PY
done
done
GOCTR_FWD_PERSIST=0 timeout 300 python bench.py --no-cpu-baseline --no-serving --steps 200 --warmup 20 > $O/np.json 2> $O/np.err
python - <<PY
import json
d=json.loads(open('$O/np.json').read().strip().splitlines()[-1]); print('nopersist', d['value'], d.get('recommend_qps'), d.get('recommend_qps_keys'))
PY
timeout 120 python scripts/dbg_predict.py 2>&1 | grep "forward-only" | tail -2
