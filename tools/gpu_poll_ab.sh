#!/bin/bash
# GPU box: does polling the stamped control block (NLBAC_CTL_POLL, odeint.py) beat the event wait?  A/B pairs in one call.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/poll; mkdir -p $O
run() {   # name, poll, bench args
  NLBAC_CTL_POLL=$2 timeout -k 10 240 python bench.py --no-cpu-baseline ${@:3} > $O/$1_$2.json 2> $O/$1_$2.err || { echo "$1 poll=$2 FAILED"; tail -5 $O/$1_$2.err; return 1; }
  python -c "import json,sys; d=json.loads(open('$O/$1_$2.json').read().strip().splitlines()[-1]); print('%-16s poll=%s  %.4f ms/update' % ('$1', '$2', d['ms_per_step']), flush=True)"
}
for rep in 1 2; do
  for poll in 0 1; do
    run pvtol_adj $poll --env Pvtol --batch 16384 --adjoint --steps 60 || exit 1
  done
done
for poll in 0 1; do run pvtol $poll --env Pvtol --batch 16384 --steps 60 || exit 1; done
for poll in 0 1; do run headline $poll --steps 100 --lean || exit 1; done
for poll in 0 1; do run cars_d5_adj $poll --env SimulatedCars --batch 8192 --solver dopri5 --adjoint --steps 60 || exit 1; done
