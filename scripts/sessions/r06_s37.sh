#!/bin/bash
# round 6, session 37: wavefront priorities (s_setprio) where wavefronts of different roles share a SIMD -- the weight-gradient launch's
# multiplying / staging wavefronts, the chain launch's second-dispatched half, ctr_fwd4's two workgroups per CU -- and the chain's output-unit
# weights requested in front of barrier (2).  GOCTR_EXP_PRIO = 10 * chain + tn (experiment knob, not in the tree afterwards);
# libgoctr_hip_b.so = -DCX_W2_EARLY=1 -DF4_PRIO=1 (priority 1 around every 6-MFMA group), _c.so = -DCX_W2_EARLY=1 -DF4_PRIO=2 (priority 1 from F0 to the exchange)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s37; mkdir -p $O
GOCTR_LIB=$R/goctr_amd/libgoctr_hip_c.so timeout 600 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_pipeline.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
train() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], {k:v.get('avg_us') for k,v in d.get('kernels',{}).items() if isinstance(v,dict)})
P
}
pred() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 100 --warmup 20 --no-cpu-baseline --no-serving --no-roofline --phase predict > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', 'qps', d.get('recommend_qps'))
P
}
for rep in 1 2 3; do
train base_$rep
train tn1_$rep GOCTR_EXP_PRIO=1
train tn2_$rep GOCTR_EXP_PRIO=2
train ch1_$rep GOCTR_EXP_PRIO=10
train ch2_$rep GOCTR_EXP_PRIO=20
train w2early_$rep GOCTR_LIB=$R/goctr_amd/libgoctr_hip_c.so
done
for rep in 1 2 3; do
pred p_base_$rep ""
pred p_b_$rep "" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_b.so
pred p_c_$rep "" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_c.so
done
pred y_base "--workload youtube"
pred y_b "--workload youtube" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_b.so
pred y_c "--workload youtube" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_c.so
