#!/bin/bash
# round 6, closing session 6 (the tree with forty switches): the whole -m gpu suite, smoke, the driver's line and the default line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_final6; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/din_steps20.json 2> $O/din_steps20.err
timeout 400 python bench.py > $O/din.json 2> $O/din.err
python - <<'P'
import json
for n in ("din_steps20", "din"):
    d = json.loads(open(f"gpurun_out/r06_final6/{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d.get("recommend_qps"), d["roofline"]["frac"], d.get("step_traffic_ratio"), d["cpu_baseline"]["value"] if "cpu_baseline" in d else None)
P
