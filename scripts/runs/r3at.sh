#!/bin/bash
O=gpurun_out/r3at; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for Z in 4096 0; do
GOCTR_SERVE_ZEROCOPY=$Z rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_$Z -- $GRAFT_REPO_ROOT/goctr_amd/host/rank_bench --threads 1 --n 256 --seconds 0.2 --coalesce 1 > $GRAFT_REPO_ROOT/$O/rb_$Z.json 2> $GRAFT_REPO_ROOT/$O/kt_$Z.err
echo "zerocopy=$Z"; head -3 $(ls $GRAFT_REPO_ROOT/$O/kt_$Z/*/*_kernel_stats.csv | head -1) | cut -c1-150
python3 -c "
import json; d=json.load(open('$GRAFT_REPO_ROOT/$O/rb_$Z.json')); print([(r['n'], r['latency_us']['p50']) for r in d['results']])"
done
