R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfgs in "din 1" "din 0" "youtube 1" "youtube 0"; do
  set -- $cfgs
  rm -rf /tmp/pq
  GOCTR_EMB_CACHE=$2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -- python $R/bench.py --workload $1 --train-emb 0.01 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /tmp/pq.json 2>/dev/null
  f=$(find /tmp/pq -name "*kernel_stats.csv" | head -1)
  echo "$1 cache=$2: grad $(grep emb_grad $f | awk -F, '{print $(NF-4)}') apply $(grep emb_apply $f | awk -F, '{print $(NF-4)}') $(cut -c60-130 /tmp/pq.json)"
done
