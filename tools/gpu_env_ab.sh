#!/bin/bash
# GPU box: A/B of one environment knob on the lean headline bench, pairs in one call.
#   bash tools/gpu_env_ab.sh <ENV_VAR> <value A> <value B> [reps] [bench args]
cd "${GRAFT_REPO_ROOT:-.}"
VAR=$1; A=$2; B=$3; REPS=${4:-2}; shift 4 || shift $#
O=gpurun_out/ab_$VAR; mkdir -p $O
for rep in $(seq 1 $REPS); do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 240 python bench.py --no-cpu-baseline --steps 100 --lean "$@" > $O/$v.json 2> $O/$v.err || { echo "$VAR=$v FAILED"; tail -5 $O/$v.err; exit 1; }
    python -c "import json; d=json.loads(open('$O/$v.json').read().strip().splitlines()[-1]); print('$VAR=$v  %.4f ms/update' % d['ms_per_step'], flush=True)"
  done
done
