R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ns in 0 2 64 2048; do
  rm -rf /tmp/pe_$ns
  GOCTR_EMB_NSLOT=$ns timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$ns -- python $R/bench.py --workload din --train-emb 0.1 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  f=$(find /tmp/pe_$ns -name "*kernel_stats.csv" | head -1)
  echo "nslot=$ns: $(grep emb_grad $f | awk -F, '{print $(NF-4)}')"
done
