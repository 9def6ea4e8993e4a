#!/bin/bash
# item2vec hot-set A/B: rows cached per LDS table (GOCTR_W2V_HOT=n) vs speed, memory-side bytes and the loss gates
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for H in 128 256; do
  echo "== GOCTR_W2V_HOT=$H"
  GOCTR_W2V_HOT=$H timeout 300 python $R/bench.py --workload item2vec --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   words/s', d['value'], 'ms/pass', d['ms_per_step'])"
  rm -rf /tmp/w2v_f /tmp/w2v_w
  GOCTR_W2V_HOT=$H rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/w2v_f -- python $R/bench.py --workload item2vec --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  GOCTR_W2V_HOT=$H rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/w2v_w -- python $R/bench.py --workload item2vec --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python3 - <<'PY'
import csv,glob
def tot(d,c):
    v=[float(r['Counter_Value']) for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True) for r in csv.DictReader(open(f)) if 'w2v_hogwild' in r['Kernel_Name'] and r['Counter_Name']==c]
    return sum(v)/max(len(v),1)
f=tot('/tmp/w2v_f','FETCH_SIZE')*1024*2; w=tot('/tmp/w2v_w','WRITE_SIZE')*1024
print('   memory-side per pass: %.1f GB = %.2f KB/word' % ((f+w)/1e9, (f+w)/1e7/1e3))
PY
done
cd $R
for H in 128 256; do
  echo "== loss gates, GOCTR_W2V_HOT=$H"
  GOCTR_W2V_HOT=$H timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_w2v.py -q -m gpu -k "item2vec or hogwild" -s 2>&1 | grep -E "HS loss|passed|failed|Error" | tail -6
done
