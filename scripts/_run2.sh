mkdir -p gpurun_out/r2
timeout 900 bash scripts/prof_workload.sh item2vec --workload item2vec > gpurun_out/r2/prof.log 2>&1; echo "prof rc=$?"
python bench.py --workload item2vec > gpurun_out/r2/bench_item2vec.json 2> gpurun_out/r2/bench.err; tail -c 600 gpurun_out/r2/bench_item2vec.json
NO_LOSS=1 bash scripts/w2v_jb.sh "0:8 3:8:2:1 4:4:4:2 4:4:8:2" > gpurun_out/r2/ab.txt 2>&1; cat gpurun_out/r2/ab.txt
