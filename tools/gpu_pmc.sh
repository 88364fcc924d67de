#!/bin/bash
# On the GPU box: HBM traffic per kernel from the TCC counters, one counter per pass
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; never combined with sys/hip tracing).
#   bash tools/gpu_pmc.sh <tag> [bench args]
# Each counter is collected over TWO lean bench processes (bench.py --lean: exactly warm-up + timed updates) that
# differ by 20 updates: per_update_bytes = (bytes of the 10 + 40 run - bytes of the 10 + 20 run) / 20, so whatever a
# process does once (the zero fills of the slot pools and arenas at allocation, packing, warm-up) cancels.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  for N in 20 40; do
    D=gpurun_out/pmc_${TAG}_${C}_$N
    rm -rf $D; mkdir -p $D
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o run -- python3 bench.py --lean --steps $N --warmup 10 "$@" > $D/bench.json 2> $D/bench.err || { tail -20 $D/bench.err; exit 1; }
    echo "pass $C x $N done"
  done
done
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_FETCH_SIZE_40 gpurun_out/pmc_${TAG}_WRITE_SIZE_40 50 \
    gpurun_out/pmc_${TAG}_FETCH_SIZE_20 gpurun_out/pmc_${TAG}_WRITE_SIZE_20 30 > gpurun_out/pmc_${TAG}_traffic.json
head -c 1500 gpurun_out/pmc_${TAG}_traffic.json
# the raw per-dispatch tables are large: keep only the summary
find gpurun_out/pmc_${TAG}_*SIZE_* -name "*.csv" -size +8M -delete || true
