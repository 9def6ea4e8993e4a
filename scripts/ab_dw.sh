# same-box A/B of the weight-gradient kernel: bf16 split (default) vs GOCTR_TN_F32=1
for i in 1 2; do
for v in 0 1; do
GOCTR_TN_F32=$v timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TN_F32=$v', d['value'], d['ms_per_step'], d['kernels']['dW0'])"
done; done
