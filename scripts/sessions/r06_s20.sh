#!/bin/bash
# round 6, session 20: phase stamps of a steady-state tile of the forward-only chain (GOCTR_DBG=chain)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
GOCTR_DBG=chain timeout 300 python scripts/ubench/fwd_phases.py 2>&1 | tail -12
