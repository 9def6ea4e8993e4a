#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
for k in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serving > $O/b20_$k.json 2> $O/b20_$k.err
  timeout 300 python bench.py --no-cpu-baseline --no-serving > $O/b200_$k.json 2> $O/b200_$k.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i/b*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})
P
rocm-smi --showclocks 2>/dev/null | head -20
