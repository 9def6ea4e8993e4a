#!/bin/bash
# round 5, GPU session 7: SQ counters of the nested vs flattened item2vec walk
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s7; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS"
run() { name=$1; shift; ( export "$@" X=1; rocprofv3 --pmc $SQ --output-format csv -d $O/$name -- python $R/bench.py --workload item2vec --no-cpu-baseline --steps 20 --warmup 20 > $O/$name.json 2> $O/$name.err )
  f=$(ls $O/$name/*/*_counter_collection.csv | head -1)
  python3 - "$f" "$name" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    if "hogwild" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_INSTS_VALU": cnt[k] += 1
for k, v in acc.items():
    n = cnt[k]
    print(sys.argv[2], k, "launches", n, {c: round(x / n / 1e9, 3) for c, x in v.items()})
PY
  find $O/$name -type f -delete
}
run flat0 GOCTR_W2V_FLAT=0
run flat1_pf8 GOCTR_W2V_FLAT=1
run flat1_pf4 GOCTR_W2V_FLAT=1 GOCTR_W2V_PF=4
