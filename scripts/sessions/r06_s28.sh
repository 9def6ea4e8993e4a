#!/bin/bash
# round 6, session 28: GPU tests of the CTR paths on the committed ctr_fwd4
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 1800 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py tests/test_gpu_fullsize.py tests/test_gpu_rank.py tests/test_gpu_model_e2e.py tests/test_gpu_resume.py -q -m gpu -p no:cacheprovider 2>&1 | tail -5
