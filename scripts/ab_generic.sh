# two default bench runs (training samples/s, ms/step, reduce kernel time)
for i in 1 2 3; do
timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels']['reduce'])"
done
