#!/bin/bash
# builds and runs scripts/ubench/bar_write.hip (GPU box): scripts/gpu.sh -- scripts/bar_write.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/bar_write; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/bar_write $R/scripts/ubench/bar_write.hip || exit 1
for cfg in "14336 977" "14336 64" "4096 977" "1024 16"; do timeout 60 /tmp/bar_write $cfg 2000; done 2>&1 | tee $O/out.txt
