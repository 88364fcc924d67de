#!/bin/bash
# GPU box: the election-free opening norms (NLBAC_NORM_DEFER, odeint.py) against the fused norms with elections.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/defer; mkdir -p $O
run() {   # name, defer, bench args
  NLBAC_NORM_DEFER=$2 timeout -k 10 240 python bench.py --no-cpu-baseline ${@:3} > $O/$1_$2.json 2> $O/$1_$2.err || { echo "$1 defer=$2 FAILED"; tail -5 $O/$1_$2.err; return 1; }
  python -c "import json,sys; d=json.loads(open('$O/$1_$2.json').read().strip().splitlines()[-1]); print('%-16s defer=%s  %.4f ms/update' % ('$1', '$2', d['ms_per_step']), flush=True)"
}
for rep in 1 2; do for v in 0 1; do run headline $v --steps 100 --lean || exit 1; done; done
for v in 0 1; do run pvtol $v --env Pvtol --batch 16384 --steps 60 || exit 1; done
