#!/bin/bash
# kernel trace of the --train-emb workloads (plan build + steps): scripts/gpu.sh -- scripts/prof_plan.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for W in din youtube; do
  OUT=$R/gpurun_out/p_${W}emb; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py --workload $W --train-emb 0.05 --steps 50 --warmup 10 --no-cpu-baseline --no-serving --no-roofline --phase train > $OUT/kt_bench.json 2> $OUT/kt.err
  f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
  (echo "== $W --train-emb: kernel stats (Name, Calls, TotalNs, AvgNs, ...)"; head -1 $f; grep -i "emb_plan\|radix\|onesweep\|scan_\|emb_slot\|emb_coef\|emb_span\|sort" $f) > $R/gpurun_out/p_${W}emb/plan_kernels.txt
  find $OUT -type f ! -name '*_kernel_stats.csv' ! -name '*.json' ! -name '*.err' ! -name '*.txt' -delete
  python $R/bench.py --workload $W --train-emb 0.05 --steps 50 --warmup 10 --no-cpu-baseline --no-serving --no-roofline --phase train > $OUT/bench.json 2> $OUT/bench.err
done
cat $R/gpurun_out/p_dinemb/plan_kernels.txt $R/gpurun_out/p_youtubeemb/plan_kernels.txt
grep -o '"plan_build[^,]*,\|"samples_per_s_incl_plan[^}]*}\|"value": [0-9.]*' $R/gpurun_out/p_dinemb/bench.json $R/gpurun_out/p_youtubeemb/bench.json
