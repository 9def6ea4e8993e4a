R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/p
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p/kt -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/gpurun_out/p/kt_bench.json 2>$R/gpurun_out/p/kt.err
GOCTR_NO_GRAPH=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p/fetch -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/p/f.json 2>$R/gpurun_out/p/f.err
GOCTR_NO_GRAPH=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p/write -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/p/w.json 2>$R/gpurun_out/p/w.err
find $R/gpurun_out/p -type f | head -30; du -sh $R/gpurun_out/p
