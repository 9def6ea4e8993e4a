#!/bin/bash
# round 3, GPU call D: plan path v3 (256-thread slot workgroups, attn_bwd fused into emb_coef)
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_embtrain.py tests/test_gpu_comm.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 30 $O/pytest.log
timeout 300 python bench.py --train-emb 0.05 --no-cpu-baseline --no-serving > $O/din_emb.json 2> $O/din_emb.err
timeout 300 python bench.py --workload youtube --train-emb 0.05 --no-cpu-baseline --no-serving > $O/yt_emb.json 2> $O/yt_emb.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3d/*_emb.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('kernels'), d.get('roofline'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
P
PASSES="kt" timeout 300 bash scripts/prof_workload.sh dinemb --train-emb 0.05 > /dev/null 2>&1
PASSES="kt" timeout 300 bash scripts/prof_workload.sh youtubeemb --workload youtube --train-emb 0.05 > /dev/null 2>&1
find gpurun_out/p_dinemb gpurun_out/p_youtubeemb -name '*_kernel_stats.csv' | xargs -I{} sh -c 'echo {}; cut -c1-110 {} | head -12'
