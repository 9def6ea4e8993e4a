#!/bin/bash
# builds and runs scripts/ubench/xcd_handoff.hip (GPU box): scripts/gpu.sh -- scripts/xcd_handoff.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/xcd_handoff; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/xcd_handoff $R/scripts/ubench/xcd_handoff.hip || exit 1
timeout 120 /tmp/xcd_handoff 200 2>&1 | tee $O/out.txt
