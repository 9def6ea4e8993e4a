#!/bin/bash
# round 6, closing session 12: rocprofv3 passes of the --train-emb lines at HEAD (their summaries were closing session 3's), then the two lines
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_final12; mkdir -p $O gpurun_out/bench
PASS_TIMEOUT=240 scripts/prof_workload.sh dinemb --train-emb 0.05 > $O/prof_dinemb.log 2>&1; tail -1 $O/prof_dinemb.log
PASS_TIMEOUT=240 scripts/prof_workload.sh youtubeemb --workload youtube --train-emb 0.05 > $O/prof_youtubeemb.log 2>&1; tail -1 $O/prof_youtubeemb.log
du -sh gpurun_out/p_dinemb gpurun_out/p_youtubeemb
