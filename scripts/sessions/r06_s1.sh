#!/bin/bash
# round 6, session 1: cross-launch XCD hand-off microbenchmark; the cfg3 line and chain / dW stamps on this round's box (baseline)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s1; mkdir -p $O
scripts/xcd_handoff.sh > $O/xcd_handoff.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/din_steps20.json 2> $O/din_steps20.err
timeout 120 python scripts/dbg_chain.py > $O/dbg_chain.txt 2>&1
cat $O/xcd_handoff.txt; tail -c 1500 $O/din_steps20.json; grep -h "phases\|tn_multi" $O/dbg_chain.txt | tail -6
