# emb_grad timing experiments on the DIN cfg3 step (GOCTR_EMB_DBG bits: 1 no flush, 2 no miss atomics, 4 no LDS adds)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for dbg in 0 7; do
  rm -rf /tmp/pe_$dbg
  GOCTR_EMB_DBG=$dbg timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$dbg -- python $R/bench.py --workload din --train-emb 0.1 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  f=$(find /tmp/pe_$dbg -name "*kernel_stats.csv" | head -1)
  echo "dbg=$dbg: $(grep emb_grad $f | awk -F, '{print $(NF-4)}')"
done
