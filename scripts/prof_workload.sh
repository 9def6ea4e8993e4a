#!/bin/bash
# rocprofv3 passes over one bench.py workload (run on the GPU box through gpurun):
#   scripts/prof_workload.sh <name> [bench.py args...]      e.g.  scripts/prof_workload.sh youtube --workload youtube
# Writes raw CSVs under gpurun_out/p_<name>/ ; scripts/prof_summarize.py <tag> gpurun_out/p_<name> turns them into profiles/.
# One PROCESS per pass and per PHASE (bench.py --phase train | predict), so that a kernel's training launches and its
# predict launches can never be mixed up -- at cfg4 they have the same grid.  Counters NEVER together with tracing, one
# counter family per pass (MI355X_MICROARCH.md "rocprofv3 PMC slots"):
#   kt     --kernel-trace --stats            per-kernel durations: graph replay as benchmarked + the eager instrumented re-run
#   fetch  --pmc FETCH_SIZE                  memory-side read bytes      (eager steps, GOCTR_NO_GRAPH=1 GOCTR_EAGER_PIPELINE=1: one
#                                            dispatch record per launch, the PIPELINED kernels of the replayed step -- reduce_attn included)
#   write  --pmc WRITE_SIZE                  memory-side write bytes
#   sq     --pmc SQ_*                        MFMA-busy, VALU / wave cycles
#   l2     --pmc TCC_HIT_sum TCC_MISS_sum    L2 hit rate
#   pkt / pfetch / pwrite / psq / pl2        the same for the predict phase (PREDICT=1; din / youtube only)
# PASSES="kt fetch write" restricts the training passes (default: all five).  Every pass runs under `timeout $PASS_TIMEOUT` (default 420 s:
# round 6 lost an hour of GPU time to ONE hung --pmc WRITE_SIZE pass of the mlp workload).  KT_EAGER=1: the kernel-trace pass with eager
# launches (GOCTR_NO_GRAPH=1) -- rocprofv3 7.2 crashes inside the graph capture of the sklearn-port MLP workloads; a kernel's duration
# does not depend on how it was launched.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
OUT=$R/gpurun_out/p_$NAME
PASSES=${PASSES:-"kt fetch write sq l2"}
TO="timeout ${PASS_TIMEOUT:-420}"
KTENV=""; [ "${KT_EAGER:-0}" = "1" ] && KTENV="env GOCTR_NO_GRAPH=1"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
(cd $R && git rev-parse --short HEAD 2>/dev/null || cat $R/.head 2>/dev/null) > $OUT/HEAD
COMMON="--no-cpu-baseline --no-serving"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY"
for P in $PASSES; do
  case $P in
    kt)    $KTENV $TO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py --steps 200 --warmup 20 $COMMON --phase train "$@" > $OUT/kt_bench.json 2> $OUT/kt.err ;;
    fetch) GOCTR_NO_GRAPH=1 GOCTR_EAGER_PIPELINE=1 $TO rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $R/bench.py --steps 30 --warmup 5 $COMMON --no-roofline --phase train "$@" > $OUT/f.json 2> $OUT/f.err ;;
    write) GOCTR_NO_GRAPH=1 GOCTR_EAGER_PIPELINE=1 $TO rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $R/bench.py --steps 30 --warmup 5 $COMMON --no-roofline --phase train "$@" > $OUT/w.json 2> $OUT/w.err ;;
    sq)    GOCTR_NO_GRAPH=1 GOCTR_EAGER_PIPELINE=1 $TO rocprofv3 --pmc $SQ --output-format csv -d $OUT/sq -- python $R/bench.py --steps 30 --warmup 5 $COMMON --no-roofline --phase train "$@" > $OUT/s.json 2> $OUT/s.err ;;
    l2)    GOCTR_NO_GRAPH=1 GOCTR_EAGER_PIPELINE=1 $TO rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python $R/bench.py --steps 30 --warmup 5 $COMMON --no-roofline --phase train "$@" > $OUT/l.json 2> $OUT/l.err ;;
  esac
done
if [ "${PREDICT:-0}" = "1" ]; then
  $TO rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pkt -- python $R/bench.py --steps 200 --warmup 20 $COMMON --no-roofline --phase predict "$@" > $OUT/pkt_bench.json 2> $OUT/pkt.err
  $TO rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pfetch -- python $R/bench.py --steps 100 --warmup 5 $COMMON --no-roofline --phase predict "$@" > $OUT/pf.json 2> $OUT/pf.err
  $TO rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pwrite -- python $R/bench.py --steps 100 --warmup 5 $COMMON --no-roofline --phase predict "$@" > $OUT/pw.json 2> $OUT/pw.err
  $TO rocprofv3 --pmc $SQ --output-format csv -d $OUT/psq -- python $R/bench.py --steps 100 --warmup 5 $COMMON --no-roofline --phase predict "$@" > $OUT/ps.json 2> $OUT/ps.err
  $TO rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pl2 -- python $R/bench.py --steps 100 --warmup 5 $COMMON --no-roofline --phase predict "$@" > $OUT/pl.json 2> $OUT/pl.err
fi
# keep only what the summariser reads (the merge back is capped at 64 MiB)
find $OUT -type f ! -name '*_kernel_stats.csv' ! -name '*_kernel_trace.csv' ! -name '*_counter_collection.csv' ! -name '*.json' ! -name '*.err' ! -name HEAD -delete
du -sh $OUT; tail -n 2 $OUT/*.err | head -40
