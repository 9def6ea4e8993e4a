#!/bin/bash
# round 6, session 22: kernel trace of the predict phase, 8-wavefront forward kernel against ctr_fwd4 (and GOCTR_FWD4 grid variants)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r06_s22; rm -rf $O; mkdir -p $O
for v in 0 1; do
  GOCTR_FWD4=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pkt$v -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving --no-roofline --phase predict > $O/pkt$v.json 2> $O/pkt$v.err
  echo "GOCTR_FWD4=$v"; cat $O/pkt$v/*/*_kernel_stats.csv | head -4 | cut -c1-150
done
GOCTR_FWD_PERSIST=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pktnp -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving --no-roofline --phase predict > $O/pktnp.json 2> $O/pktnp.err
echo "fwd4, one workgroup per tile"; cat $O/pktnp/*/*_kernel_stats.csv | head -4 | cut -c1-150
find $O -type f ! -name '*_kernel_stats.csv' ! -name '*.json' ! -name '*.err' -delete
