#!/bin/bash
# serving latency distribution of goctr_rank from 1 / 8 / 16 concurrent host threads: scripts/gpu.sh -- scripts/serve_tail.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=$R/goctr_amd/host/rank_bench
run() { echo "== $*"; env "$@" $B --threads 1,8,16 --n 32,256,2048 --seconds 0.3 --coalesce both | python3 -c "
import json,sys
d=json.load(sys.stdin)
for e in d['results']:
    l=e['latency_us']; print('n %5d t %2d coalesce %-5s calls %6d qps %7d  p50 %6.1f p90 %6.1f p99 %6.1f p999 %7.1f max %8.1f  p99/p50 %.2f' % (e['n'],e['threads'],e['coalesce'],e['calls'],round(e['rank_qps']),l['p50'],l['p90'],l['p99'],l['p999'],l['max'],l['p99']/l['p50']))
print('bit_equal_to_single_threaded', d['bit_equal_to_single_threaded'])
"; }
echo "host cores: $(nproc)"
run GOCTR_SERVE_SLOTS=8
run GOCTR_SERVE_SLOTS=4
