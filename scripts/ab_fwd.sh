# same-box A/B of the predict kernel: bf16 split (default) vs GOCTR_FWD_F32=1
for i in 1 2; do
for v in 0 1; do
GOCTR_FWD_F32=$v timeout 200 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FWD_F32=$v qps', d['recommend_qps'], 'train', d['value'])"
done; done
