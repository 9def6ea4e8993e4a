#!/bin/bash
# round 5, verification of HEAD: full -m gpu suite + smoke (no bench lines)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final3; rm -rf $O; mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/pytest.log; tail -4 $O/pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/smoke.log; cat $O/smoke.log
