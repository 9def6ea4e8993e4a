R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pm
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pm/kt -- python $R/bench.py --workload ${1:-mlp} --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/pm/bench.json 2>$R/gpurun_out/pm/err
cat $R/gpurun_out/pm/kt/*/*_kernel_stats.csv
