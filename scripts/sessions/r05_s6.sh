#!/bin/bash
# round 5, GPU session 6: the flattened item2vec walk -- parity / gates, then A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s6; rm -rf $O; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_w2v.py tests/test_gpu_corpus.py tests/test_gpu_huffman.py -m gpu -q -x 2>&1 | tail -15) > $O/pytest_w2v.log
tail -5 $O/pytest_w2v.log
B="--workload item2vec --no-cpu-baseline"
GOCTR_W2V_FLAT=0 timeout 200 python bench.py $B > $O/flat0.json 2> $O/flat0.err
GOCTR_W2V_FLAT=1 timeout 200 python bench.py $B > $O/flat1_pf8.json 2> $O/flat1_pf8.err
GOCTR_W2V_FLAT=1 GOCTR_W2V_PF=6 timeout 200 python bench.py $B > $O/flat1_pf6.json 2> $O/flat1_pf6.err
GOCTR_W2V_FLAT=1 GOCTR_W2V_PF=4 timeout 200 python bench.py $B > $O/flat1_pf4.json 2> $O/flat1_pf4.err
GOCTR_W2V_FLAT=0 GOCTR_W2V_PF=4 timeout 200 python bench.py $B > $O/flat0_pf4.json 2> $O/flat0_pf4.err
for f in $O/*.json; do python3 -c "
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['unit'], 'ms/step', d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'))
except Exception as e: print('$f', 'ERR', e)
"; done
tail -3 $O/flat1_pf8.err
(timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multi.py -m gpu -q -s -k "item2vec or cfg5 or w2v" 2>&1 | grep -v "^\.*$" | tail -15) > $O/pytest_gates.log
tail -8 $O/pytest_gates.log
