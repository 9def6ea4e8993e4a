#!/bin/bash
# rocprofv3 passes over one bench.py workload (run on the GPU box through gpurun):
#   scripts/prof_workload.sh <name> [bench.py args...]      e.g.  scripts/prof_workload.sh youtube --workload youtube
# Writes raw CSVs under gpurun_out/p_<name>/ ; scripts/prof_summarize.py <tag> gpurun_out/p_<name> turns them into profiles/.
# Passes (counters NEVER together with tracing, one counter family per pass -- MI355X_MICROARCH.md "rocprofv3 PMC slots"):
#   kt     --kernel-trace --stats            per-kernel durations (graph replay, as benchmarked)
#   fetch  --pmc FETCH_SIZE                  memory-side read bytes      (eager steps: one dispatch record per launch)
#   write  --pmc WRITE_SIZE                  memory-side write bytes
#   sq     --pmc SQ_*                        MFMA-busy, VALU / wave cycles
#   l2     --pmc TCC_HIT_sum TCC_MISS_sum    L2 hit rate
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
OUT=$R/gpurun_out/p_$NAME
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
COMMON="--no-cpu-baseline --no-serving"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py --steps 200 --warmup 20 $COMMON "$@" > $OUT/kt_bench.json 2> $OUT/kt.err
GOCTR_NO_GRAPH=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $R/bench.py --steps 30 --warmup 5 $COMMON --no-roofline "$@" > $OUT/f.json 2> $OUT/f.err
GOCTR_NO_GRAPH=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $R/bench.py --steps 30 --warmup 5 $COMMON --no-roofline "$@" > $OUT/w.json 2> $OUT/w.err
GOCTR_NO_GRAPH=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/sq -- python $R/bench.py --steps 30 --warmup 5 $COMMON --no-roofline "$@" > $OUT/s.json 2> $OUT/s.err
GOCTR_NO_GRAPH=1 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python $R/bench.py --steps 30 --warmup 5 $COMMON --no-roofline "$@" > $OUT/l.json 2> $OUT/l.err
# keep only what the summariser reads (the merge back is capped at 64 MiB)
find $OUT -type f ! -name '*_kernel_stats.csv' ! -name '*_kernel_trace.csv' ! -name '*_counter_collection.csv' ! -name '*.json' ! -name '*.err' -delete
du -sh $OUT; tail -n 2 $OUT/*.err | head -40
