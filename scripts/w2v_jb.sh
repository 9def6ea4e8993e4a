#!/bin/bash
# item2vec node-major chunks A/B: GOCTR_W2V_JB (pairs per chunk; 0 = pair-major) x GOCTR_W2V_WPS (wavefronts per SIMD) vs speed,
# memory-side bytes and the HS loss gate.  usage: scripts/w2v_jb.sh "0:8 2:8:2 4:4 6:4:8"   (JB:WPS[:PF[:CPL]])
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
VARIANTS=${1:-0:8 3:8:2:1 4:8:2:1 3:4:4:2 4:4:4:2}
for V in $VARIANTS; do
  IFS=: read JB WPS PF CPL <<< "$V"
  export GOCTR_W2V_JB=$JB GOCTR_W2V_WPS=$WPS GOCTR_W2V_PF=${PF:-2} GOCTR_W2V_CPL=${CPL:-1}
  echo "== GOCTR_W2V_JB=$JB GOCTR_W2V_WPS=$WPS GOCTR_W2V_PF=$GOCTR_W2V_PF GOCTR_W2V_CPL=$GOCTR_W2V_CPL"
  timeout 300 python $R/bench.py --workload item2vec --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   words/s', d['value'], 'ms/pass', d['ms_per_step'])"
  if [ -z "$NO_PMC" ]; then
  rm -rf /tmp/w2v_f /tmp/w2v_w
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/w2v_f -- python $R/bench.py --workload item2vec --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/w2v_w -- python $R/bench.py --workload item2vec --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python3 - <<'PY'
import csv,glob
def tot(d,c):
    v=[float(r['Counter_Value']) for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True) for r in csv.DictReader(open(f)) if 'w2v_hogwild' in r['Kernel_Name'] and r['Counter_Name']==c]
    return sum(v)/max(len(v),1)
f=tot('/tmp/w2v_f','FETCH_SIZE')*1024*2; w=tot('/tmp/w2v_w','WRITE_SIZE')*1024
print('   memory-side per pass: %.1f GB = %.2f KB/word' % ((f+w)/1e9, (f+w)/1e7/1e3))
PY
  fi
  [ -z "$NO_LOSS" ] && (cd $R && timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "cfg5_item2vec" -s 2>&1 | grep -E "HS loss|passed|failed|Error" | tail -4)
done
