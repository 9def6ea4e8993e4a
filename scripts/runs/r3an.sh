#!/bin/bash
O=gpurun_out/r3an; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py tests/test_gpu_fullsize.py tests/test_gpu_rank.py tests/test_gpu_embtrain.py tests/test_gpu_comm.py -q -m gpu -k "not item2vec" -p no:cacheprovider -x 2>&1 | tail -3
run() { # label lib args...
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
run $L ytemb --workload youtube --train-emb 0.05
done
done
for L in libgoctr_hip.so libgoctr_hip_old.so; do
LD_LIBRARY_PATH= GOCTR_LIB=$PWD/goctr_amd/$L timeout 100 python - <<PY
import subprocess,os,json,shutil
# rank_bench links libgoctr_hip.so by rpath: swap the file for the run
PY
done
