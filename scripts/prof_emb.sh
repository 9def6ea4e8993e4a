# per-kernel times of a training step with the trainable-embedding extension on (DIN cfg3 and YouTube cfg4 slices)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pe; mkdir -p $R/gpurun_out/pe
for w in din youtube; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pe/$w -- python $R/bench.py --workload $w --train-emb 0.01 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/pe/$w.json 2>$R/gpurun_out/pe/$w.err
  f=$(find $R/gpurun_out/pe/$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"; cut -d, -f1-4 $f | head -22 | cut -c1-160
done
find $R/gpurun_out/pe -name "*.csv" ! -name "*kernel_stats.csv" -delete
