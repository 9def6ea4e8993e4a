R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/p
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p/kt -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/gpurun_out/p/kt_bench.json 2>$R/gpurun_out/p/kt.err
GOCTR_NO_GRAPH=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p/fetch -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/p/f.json 2>$R/gpurun_out/p/f.err
GOCTR_NO_GRAPH=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p/write -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/p/w.json 2>$R/gpurun_out/p/w.err
GOCTR_NO_GRAPH=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/p/sq -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/p/s.json 2>$R/gpurun_out/p/s.err
find $R/gpurun_out/p -type f | head -30; du -sh $R/gpurun_out/p
