#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace stats of a short bench run.  Usage: bash tools/gpu_prof.sh <tag> [bench args]
set -e
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o run -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/prof_$TAG/bench.json 2> gpurun_out/prof_$TAG/bench.err || { tail -20 gpurun_out/prof_$TAG/bench.err; exit 1; }
cat gpurun_out/prof_$TAG/bench.json
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -30 {}'
# keep only the summaries (traces are large)
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete || true
