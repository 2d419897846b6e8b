#!/bin/bash
# build everything in-tree (library, OFX bundles, oracle, mock host), then run a command on the GPU box through gpurun
# usage: tools/gpu.sh [--timeout S] '<command>'
cd "$(dirname "$0")/.." || exit 1
python -c "import __graft_entry__ as g; g.build()" || exit 1
T=900
if [ "$1" = "--timeout" ]; then T=$2; shift 2; fi
/usr/local/graft/bin/gpurun --timeout $T -- "mkdir -p gpurun_out/r6; $*"
