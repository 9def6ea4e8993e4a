#!/bin/bash
O=gpurun_out/r3w; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/din_driver.json 2> $O/din_driver.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3w/din_driver.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:1200]); print({k:d[k] for k in d if k.startswith('rank_') or k.startswith('recommend')})
P
