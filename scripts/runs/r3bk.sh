#!/bin/bash
O=gpurun_out/r3bk; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-serving --no-roofline --phase train --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$O/b.json 2> $GRAFT_REPO_ROOT/$O/kt.err
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/r3bk/kt/*/*_kernel_trace.csv')[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:60]) for r in csv.DictReader(open(f))]
rows.sort()
# the last 70 kernels: the timed call (20 steps x 3 + few)
tail=rows[-75:]
t0=tail[0][0]
prev=None
for s,e,n in tail:
    gap = (s-prev)/1e3 if prev else 0
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:6.1f}  gap {gap:6.1f}  {n}")
    prev=e
P
