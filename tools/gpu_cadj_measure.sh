#!/bin/bash
# The single-net continuous adjoint: one nlbac_concat_adj_step launch per attempted step against the stage-by-stage
# launches (NLBAC_CONCAT_ADJ_RR=0), and the direct (discrete) backward beside them.  One gpurun call.
O=gpurun_out/r4/cadj
mkdir -p $O
for solver in rk4 dopri5; do
  python bench.py --env SimulatedCars --batch 8192 --solver $solver --steps 60 --no-cpu-baseline > $O/cars_${solver}_direct.json 2> $O/cars_${solver}_direct.err &&
  python bench.py --env SimulatedCars --batch 8192 --solver $solver --steps 60 --no-cpu-baseline --adjoint > $O/cars_${solver}_adjoint.json 2> $O/cars_${solver}_adjoint.err &&
  NLBAC_CONCAT_ADJ_RR=0 python bench.py --env SimulatedCars --batch 8192 --solver $solver --steps 60 --no-cpu-baseline --adjoint > $O/cars_${solver}_adjoint_staged.json 2> $O/cars_${solver}_adjoint_staged.err || exit 1
done
python bench.py --env QuadrotorLike --batch 32768 --steps 40 --no-cpu-baseline --adjoint > $O/quad_adjoint.json 2> $O/quad_adjoint.err &&
NLBAC_CONCAT_ADJ_RR=0 python bench.py --env QuadrotorLike --batch 32768 --steps 40 --no-cpu-baseline --adjoint > $O/quad_adjoint_staged.json 2> $O/quad_adjoint_staged.err
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("%-40s %.3f ms  %.2f M/s  kernel %s frac %.3f  update frac %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"] / 1e6,
      r.get("kernel"), r.get("frac", 0), (r.get("update") or {}).get("frac")))
PY
done
