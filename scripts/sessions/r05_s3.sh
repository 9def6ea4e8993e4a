#!/bin/bash
# round 5, GPU session 3: kernel traces of the forked / unforked DIN + YouTube steps, k-NN fold without fences, item2vec DP sweep, full suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-serving --no-roofline --phase train"
trace() {  # name env... -- args
  local name=$1; shift
  ( export "$@" GOCTR_X=1; rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -- python $R/bench.py --steps 200 --warmup 20 $B $EXTRA > $O/kt_$name.json 2> $O/kt_$name.err )
  f=$(ls $O/kt_$name/*/*_kernel_stats.csv 2>/dev/null | head -1)
  echo "== $name: $(python3 -c "import json;d=json.loads(open('$O/kt_$name.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'])" 2>/dev/null)"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("   %-60s calls %6s avg %9.2f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  find $O/kt_$name -type f ! -name '*_kernel_stats.csv' -delete
}
EXTRA=""
trace din_fork1 GOCTR_FORK_ATTN=1
trace din_fork0 GOCTR_FORK_ATTN=0
trace din_r4 GOCTR_FORK_ATTN=0 GOCTR_ATT0_EARLY=0
EXTRA="--workload youtube"
trace yt_fork1 GOCTR_FORK_ATTN=1
trace yt_fork1_fat GOCTR_FORK_ATTN=1 GOCTR_ATTN_LEAN=0
trace yt_fork0 GOCTR_FORK_ATTN=0
cd $R
BB="--no-cpu-baseline --no-serving --no-roofline"
for f in 1 0; do
  GOCTR_FORK_ATTN=$f timeout 200 python bench.py $BB > $O/din_fork$f.json 2> $O/din_fork$f.err
  GOCTR_FORK_ATTN=$f timeout 300 python bench.py $BB --workload youtube > $O/yt_fork$f.json 2> $O/yt_fork$f.err
done
GOCTR_FORK_ATTN=0 GOCTR_ATT0_EARLY=0 timeout 200 python bench.py $BB > $O/din_r4path.json 2> $O/din_r4path.err
for g in 8 16; do GOCTR_KNN_G=$g timeout 200 python bench.py --workload knn --no-cpu-baseline > $O/knn_g$g.json 2> $O/knn_g$g.err; done
GOCTR_KNN_FOLD=0 timeout 200 python bench.py --workload knn --no-cpu-baseline > $O/knn_nofold.json 2> $O/knn_nofold.err
for f in $O/*.json; do python3 -c "
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['unit'], 'ms/step', d.get('ms_per_step'), 'spread', d.get('timed_region_spread'))
except Exception as e: print('$f', 'ERR', e)
"; done
timeout 600 python scripts/w2v_dp_gpu_sweep.py > $O/w2v_dp_sweep.txt 2> $O/w2v_dp_sweep.err; cat $O/w2v_dp_sweep.txt; tail -3 $O/w2v_dp_sweep.err
(timeout 1300 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize.py::test_cfg5_item2vec_w8_exchange_cadence_vs_oracle 2>&1 | tail -30) > $O/pytest.log
tail -8 $O/pytest.log
