#!/bin/bash
# round 6: rocprofv3 passes of every workload at HEAD (scripts/prof_workload.sh) + every bench line un-profiled (scripts/bench_round.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_prof1; mkdir -p $O
PREDICT=1 scripts/prof_workload.sh din > $O/prof_din.log 2>&1
PREDICT=1 scripts/prof_workload.sh youtube --workload youtube > $O/prof_youtube.log 2>&1
scripts/prof_workload.sh mlp --workload mlp > $O/prof_mlp.log 2>&1
scripts/prof_workload.sh knn --workload knn > $O/prof_knn.log 2>&1
scripts/prof_workload.sh item2vec --workload item2vec > $O/prof_item2vec.log 2>&1
scripts/prof_workload.sh dinemb --train-emb 0.05 > $O/prof_dinemb.log 2>&1
scripts/prof_workload.sh youtubeemb --workload youtube --train-emb 0.05 > $O/prof_youtubeemb.log 2>&1
GOCTR_NO_GRAPH=1 PASSES=kt scripts/prof_workload.sh mlp100k --workload mlp100k --regions 1 > $O/prof_mlp100k.log 2>&1
tail -n 3 $O/prof_*.log
du -sh gpurun_out/p_*
