#!/bin/bash
# round 6, session 13: the whole -m gpu suite on HEAD (pruned switches, the new equivalence tests, the segmented item2vec exchange test);
# smoke; the k-NN line with the compiled host
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s13; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -25 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 300 python bench.py --workload knn --steps 200 --warmup 20 > $O/knn.json 2> $O/knn.err; tail -c 1800 $O/knn.json
