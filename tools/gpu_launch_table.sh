#!/bin/bash
# On the GPU box: kernel launches per update of every task, from rocprofv3 kernel traces of two lean bench runs that
# differ by 20 updates (what the extra 20 updates launched / 20; includes 2 of the NODE fits that come every 10 updates).
#   bash tools/gpu_launch_table.sh <tag> [extra bench args]     ->  gpurun_out/<tag>_launches.txt
set -e
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_launches.txt
: > $OUT
for ENV in Unicycle UnicycleBarrier SimulatedCars Pvtol PvtolBarrier QuadrotorLike; do
  for N in 20 40; do
    D=gpurun_out/lc_${TAG}_${ENV}_$N
    rm -rf $D; mkdir -p $D
    rocprofv3 --kernel-trace --stats --output-format csv -d $D -o run -- python3 bench.py --lean --no-cpu-baseline --env $ENV --steps $N --warmup 10 "$@" > $D/bench.json 2> $D/bench.err || { tail -5 $D/bench.err; exit 1; }
    find $D -name "*kernel_trace.csv" -delete || true
  done
  python3 - $ENV gpurun_out/lc_${TAG}_${ENV}_20 gpurun_out/lc_${TAG}_${ENV}_40 >> $OUT <<'PY'
import csv, glob, sys
env, d20, d40 = sys.argv[1:4]
def calls(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = calls(d20), calls(d40)
names = sorted(b, key=lambda k: -(b[k][1] - a.get(k, (0, 0))[1]))
tot = sum(b[k][0] - a.get(k, (0, 0))[0] for k in names) / 20.0
us = sum(b[k][1] - a.get(k, (0, 0))[1] for k in names) / 20.0 / 1e3
print("%-16s %6.1f kernel launches / update   %8.1f us of kernel time / update" % (env, tot, us))
for k in names[:8]:
    print("      %6.2f x  %8.1f us   %s" % ((b[k][0] - a.get(k, (0, 0))[0]) / 20.0, (b[k][1] - a.get(k, (0, 0))[1]) / 20.0 / 1e3, k[:90]))
PY
done
cat $OUT
