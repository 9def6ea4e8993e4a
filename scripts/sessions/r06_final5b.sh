#!/bin/bash
# round 6, closing session 5, part 2: every bench line (reading part 1's r06_din_kernels.json / r06_youtube_kernels.json), the whole -m gpu
# suite, smoke
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_final5; mkdir -p $O
scripts/bench_round.sh 2>&1 | tail -24
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
