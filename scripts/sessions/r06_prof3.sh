#!/bin/bash
# round 6: what r06_prof2.sh left: the mlp counter passes (4000 eager warm-up steps under --pmc were the "hang"), the --train-emb lines
# on the tree without the chain-head experiment (kernel symbols), predict-only write-through h0 A/B, every bench line un-profiled
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_prof3; mkdir -p $O
for rep in 1 2 3; do for x in 0 1; do
GOCTR_PRED_WT=$x timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-serving --no-roofline > $O/pw${x}_$rep.json 2> $O/pw${x}_$rep.err
python - <<P
import json
d=json.loads(open('$O/pw${x}_$rep.json').read().strip().splitlines()[-1]); print('pred_wt=$x rep $rep qps', d.get('recommend_qps'), 'train', d['value'])
P
done; done
GOCTR_PRED_WT=1 timeout 300 python bench.py --workload youtube --steps 50 --warmup 10 --no-cpu-baseline --no-serving --no-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('youtube pred_wt=1 qps', d['recommend_qps'])"
GOCTR_PRED_WT=0 timeout 300 python bench.py --workload youtube --steps 50 --warmup 10 --no-cpu-baseline --no-serving --no-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('youtube pred_wt=0 qps', d['recommend_qps'])"
KT_EAGER=1 PASS_TIMEOUT=300 scripts/prof_workload.sh mlp --workload mlp > $O/prof_mlp.log 2>&1
PASS_TIMEOUT=300 scripts/prof_workload.sh dinemb --train-emb 0.05 > $O/prof_dinemb.log 2>&1
PASS_TIMEOUT=300 scripts/prof_workload.sh youtubeemb --workload youtube --train-emb 0.05 > $O/prof_youtubeemb.log 2>&1
du -sh gpurun_out/p_mlp gpurun_out/p_dinemb gpurun_out/p_youtubeemb
scripts/bench_round.sh 2>&1 | tail -30
