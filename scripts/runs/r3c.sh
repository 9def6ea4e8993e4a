#!/bin/bash
# round 3, GPU call C: plan path v2 (x carried, W0pvT by Adam), fixed-size exchange, ballot assembly
O=gpurun_out/r3c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_embtrain.py tests/test_gpu_comm.py tests/test_gpu_rank.py tests/test_gpu_assembly.py tests/test_gpu_resume.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 30 $O/pytest.log
timeout 300 python bench.py --train-emb 0.05 --no-cpu-baseline --no-serving > $O/din_emb.json 2> $O/din_emb.err
timeout 300 python bench.py --workload youtube --train-emb 0.05 --no-cpu-baseline --no-serving > $O/yt_emb.json 2> $O/yt_emb.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c/*_emb.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('kernels'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
P
timeout 120 ./goctr_amd/host/rank_bench --threads 1,8 --n 32,256,2048 --seconds 0.4 --coalesce 1 > $O/rank.json 2> $O/rank.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r3c/rank.json'))
print([(e['n'],e['threads'],round(e['rank_qps']),e['latency_us']['p50']) for e in d['results']], d['bit_equal_to_single_threaded'])
P
cd /tmp && export TMPDIR=/tmp
for N in 50 100; do
  rocprofv3 --hip-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/hip$N -- python $GRAFT_REPO_ROOT/scripts/dp_emb_trace.py $N > $GRAFT_REPO_ROOT/$O/hip$N.out 2> $GRAFT_REPO_ROOT/$O/hip$N.err
done
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv,glob,json
t={}
for N in (50,100):
    f=glob.glob(f'gpurun_out/r3c/hip{N}/**/*hip_api_stats.csv', recursive=True)
    if not f: print('no hip stats for', N, open(f'gpurun_out/r3c/hip{N}.err').read()[-600:]); continue
    t[N]={r['Name']:int(r['Calls']) for r in csv.DictReader(open(f[0]))}
if len(t)==2:
    d={k:t[100].get(k,0)-t[50].get(k,0) for k in set(t[50])|set(t[100])}
    d={k:v for k,v in d.items() if v}
    print('HIP API calls of 50 extra data-parallel train-emb steps:', json.dumps(d, sort_keys=True))
    json.dump({"what":"rocprofv3 --hip-trace --stats of scripts/dp_emb_trace.py with 100 and with 50 steps (one-rank RCCL communicator, DIN cfg3 shapes, trainable embeddings, fixed-size sparse exchange): per-API call count difference = the host-side calls of 50 steps","calls_50_steps":t[50],"calls_100_steps":t[100],"difference":d}, open('gpurun_out/r3c/hip_api_diff.json','w'), indent=1)
P
find $O -name '*.csv' -size +2M -delete
PASSES="kt" timeout 300 bash scripts/prof_workload.sh dinemb --train-emb 0.05 > /dev/null 2>&1
PASSES="kt" timeout 300 bash scripts/prof_workload.sh youtubeemb --workload youtube --train-emb 0.05 > /dev/null 2>&1
find gpurun_out/p_dinemb gpurun_out/p_youtubeemb -name '*_kernel_stats.csv' | xargs -I{} sh -c 'echo {}; cut -c1-120 {} | head -12'
