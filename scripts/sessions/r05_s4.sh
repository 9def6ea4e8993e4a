#!/bin/bash
# round 5, GPU session 4: att0_step fixed -> DIN forked pipeline A/B + trace; k-NN fold / G; item2vec DP gate; full suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s4; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-serving --no-roofline --phase train"
trace() {
  local name=$1; shift
  ( export "$@" GOCTR_X=1; rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -- python $R/bench.py --steps 200 --warmup 20 $B $EXTRA > $O/kt_$name.json 2> $O/kt_$name.err )
  f=$(ls $O/kt_$name/*/*_kernel_stats.csv 2>/dev/null | head -1)
  echo "== $name: $(python3 -c "import json;d=json.loads(open('$O/kt_$name.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'])" 2>/dev/null)"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    print("   %-60s calls %6s avg %9.2f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  find $O/kt_$name -type f ! -name '*_kernel_stats.csv' -delete
}
EXTRA=""
trace din_fork1 GOCTR_FORK_ATTN=1
trace din_fork0 GOCTR_FORK_ATTN=0
cd $R
BB="--no-cpu-baseline --no-serving --no-roofline"
for f in 1 0; do
  GOCTR_FORK_ATTN=$f timeout 200 python bench.py $BB > $O/din_fork$f.json 2> $O/din_fork$f.err
  GOCTR_FORK_ATTN=$f timeout 200 python bench.py $BB --steps 20 --warmup 5 > $O/din20_fork$f.json 2> $O/din20_fork$f.err
done
GOCTR_FORK_ATTN=0 GOCTR_ATT0_EARLY=0 timeout 200 python bench.py $BB > $O/din_r4path.json 2> $O/din_r4path.err
GOCTR_FORK_ATTN=0 GOCTR_ATT0_EARLY=0 timeout 200 python bench.py $BB --steps 20 --warmup 5 > $O/din20_r4path.json 2> $O/din20_r4path.err
timeout 300 python bench.py $BB --workload youtube > $O/yt.json 2> $O/yt.err
for g in 8 16; do GOCTR_KNN_G=$g timeout 200 python bench.py --workload knn --no-cpu-baseline > $O/knn_g$g.json 2> $O/knn_g$g.err; done
GOCTR_KNN_FOLD=0 timeout 200 python bench.py --workload knn --no-cpu-baseline > $O/knn_nofold.json 2> $O/knn_nofold.err
GOCTR_KNN_FOLD=0 GOCTR_KNN_G=16 timeout 200 python bench.py --workload knn --no-cpu-baseline > $O/knn_nofold_g16.json 2> $O/knn_nofold_g16.err
for f in $O/*.json; do python3 -c "
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['unit'], 'ms/step', d.get('ms_per_step'), 'spread', d.get('timed_region_spread'))
except Exception as e: print('$f', 'ERR', e)
"; done
(timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^\.*$" | tail -40) > $O/pytest.log
tail -12 $O/pytest.log
