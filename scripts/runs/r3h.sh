#!/bin/bash
# round 3, GPU call H: whole -m gpu suite + smoke + the driver's bench line at HEAD
O=gpurun_out/r3h; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu --timeout 800 -p no:cacheprovider --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 30 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; tail -c 300 $O/bench20.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3h/bench20.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['recommend_qps'], d.get('recommend_qps_keys'), d['roofline'], d.get('gather_roofline'))
print({k:v for k,v in d.items() if k.startswith('rank_') and k!='rank_note'})
P
