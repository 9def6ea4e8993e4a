#!/bin/bash
# round 6, session 23: ctr_fwd4 with 1 / 2 / 4 persistent workgroups per CU (is the second workgroup resident?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r06_s23; rm -rf $O; mkdir -p $O
for v in 1 2 4; do
  GOCTR_FWD4_WGS=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pkt$v -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving --no-roofline --phase predict > $O/pkt$v.json 2> $O/pkt$v.err
  echo "GOCTR_FWD4_WGS=$v"; cat $O/pkt$v/*/*_kernel_stats.csv | head -3 | cut -c1-150
  head -1 $O/pkt$v/*/*_kernel_trace.csv; grep fwd4 $O/pkt$v/*/*_kernel_trace.csv | sed -n 50,52p
  GOCTR_FWD4_WGS=$v GOCTR_DBG=chain timeout 300 python $R/scripts/ubench/fwd_phases.py 2>&1 | tail -2 | head -1
done
find $O -type f ! -name '*_kernel_stats.csv' ! -name '*.json' ! -name '*.err' -delete
