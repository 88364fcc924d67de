#!/bin/bash
# GPU box: the headline bench against several builds of the library, one after another in ONE call:
#   bash tools/ab_bench.sh "<bench args>" base <variant> <variant> ...     (base = the product library)
# prints ms/update and the event-timed per-entry-point averages of each; full lines under gpurun_out/ab/
cd "${GRAFT_REPO_ROOT:-.}"
ARGS=$1; shift
mkdir -p gpurun_out/ab
LIBDIR=$(echo neural-*-nlbac_amd/lib)
for v in "$@"; do
  if [ "$v" = base ]; then unset NLBAC_HIP_LIB; else export NLBAC_HIP_LIB=$PWD/$LIBDIR/variants/libnlbac_hip_$v.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline $ARGS > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err || { echo "$v FAILED"; tail -3 gpurun_out/ab/$v.err; continue; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open("gpurun_out/ab/%s.json" % v).read().strip().splitlines()[-1])
a = d["roofline"].get("all", {})
print("%-14s %.4f ms/update  pipelined %.4f | " % (v, d["ms_per_step"], d["pipelined"]["ms_per_step"]) +
      "  ".join("%s %.1f" % (k.replace("nlbac_", ""), x["avg_us"]) for k, x in a.items()), flush=True)
PY
done
