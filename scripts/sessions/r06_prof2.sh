#!/bin/bash
# round 6: the rocprofv3 passes r06_prof1.sh did not reach (its mlp WRITE_SIZE pass hung until the session's limit): mlp (kernel trace
# eager: rocprofv3 crashes inside this workload's graph capture), knn, item2vec, the --train-emb lines
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_prof2; mkdir -p $O
scripts/prof_workload.sh knn --workload knn > $O/prof_knn.log 2>&1
scripts/prof_workload.sh item2vec --workload item2vec > $O/prof_item2vec.log 2>&1
scripts/prof_workload.sh dinemb --train-emb 0.05 > $O/prof_dinemb.log 2>&1
scripts/prof_workload.sh youtubeemb --workload youtube --train-emb 0.05 > $O/prof_youtubeemb.log 2>&1
KT_EAGER=1 PASS_TIMEOUT=240 scripts/prof_workload.sh mlp --workload mlp > $O/prof_mlp.log 2>&1
tail -n 3 $O/prof_*.log | cut -c1-200
du -sh gpurun_out/p_*
