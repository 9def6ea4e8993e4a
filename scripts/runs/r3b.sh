#!/bin/bash
# round 3, GPU call B: plan-path sparse update (tests, A/B benches, kernel traces) + zero-copy serving A/B
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_embtrain.py tests/test_gpu_comm.py tests/test_gpu_rank.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 30 $O/pytest.log
for P in 1 0; do
  GOCTR_EMB_PLAN=$P timeout 300 python bench.py --train-emb 0.05 --no-cpu-baseline --no-serving > $O/din_emb_plan$P.json 2> $O/din_emb_plan$P.err
  GOCTR_EMB_PLAN=$P timeout 300 python bench.py --workload youtube --train-emb 0.05 --no-cpu-baseline --no-serving > $O/yt_emb_plan$P.json 2> $O/yt_emb_plan$P.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3b/*emb_plan*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('kernels'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
P
timeout 120 ./goctr_amd/host/rank_bench --threads 1,8 --n 32,256,2048 --seconds 0.4 --coalesce 1 > $O/rank_zc.json 2> $O/rank_zc.err
GOCTR_SERVE_ZEROCOPY=0 timeout 120 ./goctr_amd/host/rank_bench --threads 1,8 --n 32,256,2048 --seconds 0.4 --coalesce 1 > $O/rank_dma.json 2> $O/rank_dma.err
python - <<'P'
import json
for f in ('rank_zc','rank_dma'):
    d=json.load(open(f'gpurun_out/r3b/{f}.json'))
    print(f, [(e['n'],e['threads'],round(e['rank_qps']),e['latency_us']['p50']) for e in d['results']], d['bit_equal_to_single_threaded'])
P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/rank_kt -- $GRAFT_REPO_ROOT/goctr_amd/host/rank_bench --threads 1 --n 256 --seconds 0.2 --coalesce 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/rank_kt.err
cd $GRAFT_REPO_ROOT
find $O/rank_kt -name '*_kernel_stats.csv' | head -1 | xargs cut -c1-160 | head -12
PASSES="kt" timeout 300 bash scripts/prof_workload.sh dinemb --train-emb 0.05 > /dev/null 2>&1
PASSES="kt" timeout 300 bash scripts/prof_workload.sh youtubeemb --workload youtube --train-emb 0.05 > /dev/null 2>&1
find gpurun_out/p_dinemb gpurun_out/p_youtubeemb -name '*_kernel_stats.csv' | xargs -I{} sh -c 'echo {}; cut -c1-150 {} | head -14'
