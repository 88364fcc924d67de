#!/bin/bash
# On the GPU box: bench (+ optional rocprofv3 kernel stats).  Usage: bash tools/gpu_bench.sh [bench args]
set -e
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/smoke.log
timeout -k 10 900 python bench.py "$@" 2> gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
