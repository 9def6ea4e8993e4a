#!/bin/bash
# round 3, GPU call A: new parity tests + serving + DIN profile with the per-phase summariser
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 40 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; tail -c 600 $O/bench20.err
timeout 300 python bench.py > $O/bench200.json 2> $O/bench200.err
timeout 120 ./goctr_amd/host/rank_bench --threads 1,2,8 --n 32,256,2048 --seconds 0.5 > $O/rank.json 2> $O/rank.err; cat $O/rank.json | head -c 3000
PREDICT=1 timeout 600 bash scripts/prof_workload.sh din
