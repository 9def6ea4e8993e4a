#!/bin/bash
O=gpurun_out/r3q; mkdir -p $O
for K in 3 5; do
GOCTR_MLP_TN_KTW=$K timeout 300 python bench.py --workload mlp --no-cpu-baseline > $O/mlp_ktw$K.json 2> $O/mlp_ktw$K.err
GOCTR_MLP_TN_KTW=$K timeout 300 python bench.py --workload mlp --no-cpu-baseline > $O/mlp_ktw${K}_b.json 2> $O/mlp_ktw${K}_b.err
done
GOCTR_MLP_TN_KTW=5 timeout 600 python -m pytest tests/test_gpu_mlp.py "tests/test_gpu_fullsize.py::test_cfg2_mlp_full_size_step_vs_oracle" -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest_ktw5.log 2>&1; tail -n 3 $O/pytest_ktw5.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3q/mlp*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
P
cd /tmp && export TMPDIR=/tmp
GOCTR_MLP_TN_KTW=5 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt5 -- python $GRAFT_REPO_ROOT/bench.py --workload mlp --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $O/kt5 -name '*_kernel_stats.csv' | head -1 | xargs cut -c1-100 | head -6
find $O/kt5 -name '*_kernel_trace.csv' -delete
