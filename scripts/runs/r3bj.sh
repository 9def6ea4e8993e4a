#!/bin/bash
O=gpurun_out/r3bj; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu --timeout 800 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 4 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_din_steps20.json 2> $O/bench_din_steps20.err
timeout 300 python bench.py > $O/bench_din.json 2> $O/bench_din.err
python - <<'P'
import json
for f in ('bench_din_steps20','bench_din'):
    d=json.loads(open(f'gpurun_out/r3bj/{f}.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('recommend_qps'), d.get('recommend_qps_keys'), d['rank_latency_us'])
P
