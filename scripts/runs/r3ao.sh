#!/bin/bash
O=gpurun_out/r3ao; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py tests/test_gpu_fullsize.py tests/test_gpu_rank.py tests/test_gpu_embtrain.py -q -m gpu -k "not item2vec" -p no:cacheprovider -x 2>&1 | tail -3
run() { # lib label args...
  L=$1; shift; N=$1; shift
  GOCTR_LIB=$PWD/goctr_amd/$L timeout 300 python bench.py --no-cpu-baseline --no-serving --steps 200 --warmup 20 "$@" > $O/x.json 2> $O/x.err
  python - <<PY
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('$N $L', round(d['value']/1e6,1), d['ms_per_step'], 'qps', round((d.get('recommend_qps') or 0)/1e6,1), {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
PY
}
for R in 1 2; do
for L in libgoctr_hip.so libgoctr_hip_old.so; do
run $L din
run $L youtube --workload youtube
run $L dinemb --train-emb 0.05
done
done
for G in 4 8; do
GOCTR_PRED_GROUP=$G timeout 300 python bench.py --no-cpu-baseline --no-serving --steps 200 --warmup 20 > $O/g$G.json 2> $O/g$G.err
python - <<PY
import json
d=json.loads(open('$O/g$G.json').read().strip().splitlines()[-1]); print('group=$G', d['value'], d.get('recommend_qps'))
PY
done
timeout 120 python scripts/dbg_predict.py 2>&1 | grep "forward-only" | tail -2
