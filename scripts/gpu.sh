#!/bin/bash
# scripts/gpu.sh [--timeout S] -- <command>: stamp the snapshot with HEAD (.git does not travel) and run it on the GPU box
cd "$(dirname "$0")/.." && git rev-parse --short HEAD > .head
exec /usr/local/graft/bin/gpurun "$@"
