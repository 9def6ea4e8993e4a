#!/bin/bash
# round 6, session 3: dW / chain stamps with and without the XCD-affine dealing
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s3; mkdir -p $O
for x in 0 1 0 1; do echo "affine=$x"; GOCTR_XCD_AFFINE=$x timeout 120 python scripts/dbg_chain.py > $O/dbg_x$x.txt 2>&1; grep -h "phases\|dW x3" $O/dbg_x$x.txt | tail -8; done
