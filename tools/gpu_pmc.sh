#!/bin/bash
# On the GPU box: HBM traffic per kernel from the TCC counters, one counter per pass
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; never combined with sys/hip tracing).
#   bash tools/gpu_pmc.sh <tag> [bench args]      (the bench runs --lean --steps 30 --warmup 10: exactly 40 updates)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  D=gpurun_out/pmc_${TAG}_$C
  mkdir -p $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o run -- python3 bench.py --lean --steps 30 --warmup 10 "$@" > $D/bench.json 2> $D/bench.err || { tail -20 $D/bench.err; exit 1; }
  echo "pass $C done"
done
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE 40 > gpurun_out/pmc_${TAG}_traffic.json
head -c 1500 gpurun_out/pmc_${TAG}_traffic.json
# the raw per-dispatch tables are large: keep only the summary
find gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*.csv" -size +8M -delete || true
