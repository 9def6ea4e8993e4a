#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for M in 8 32 64; do
  echo "== GOCTR_W2V_MERGE=$M"
  GOCTR_W2V_MERGE=$M timeout 300 python bench.py --workload item2vec --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   words/s', d['value'])"
  GOCTR_W2V_MERGE=$M timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "cfg5_item2vec" -s 2>&1 | grep -E "HS loss|passed|failed"
done
