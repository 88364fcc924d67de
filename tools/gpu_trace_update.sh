#!/bin/bash
# GPU box: kernel trace of a short lean headline run; prints one update's launches in order (name, start offset, duration)
# and the per-update table.   bash tools/gpu_trace_update.sh <tag> [bench args]
set -e
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
D=gpurun_out/trace_$TAG; rm -rf $D; mkdir -p $D
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $D -o run -- python3 bench.py --lean --no-cpu-baseline --steps 60 --warmup 20 "$@" > $D/bench.json 2> $D/bench.err || { tail -5 $D/bench.err; exit 1; }
F=$(find $D -name "*kernel_trace.csv" | head -1)
python3 tools/update_table.py $F 61 69 > $D/table.txt
python3 tools/gap_report.py $F 61 69 > $D/gap_report.txt 2>&1 || true
python3 - $F > $D/one_update.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]) for r in rows)
marks = [i for i, k in enumerate(ks) if k[2].startswith("sample_rows")]
import os
want = [int(x) for x in os.environ.get("TRACE_UPDATES", "63,64").split(",")]
if os.environ.get("TRACE_FIND"):       # the first segment behind mark 40 that holds a kernel of this name (e.g. a NODE fit's)
    want = [u for u in range(40, len(marks) - 1) if any(os.environ["TRACE_FIND"] in k[2] for k in ks[marks[u]:marks[u + 1]])][:1]
for u in want:
    seg = ks[marks[u]:marks[u + 1]]
    t0 = seg[0][0]
    prev = t0
    for s, e, n in seg:
        print("%8.1f  gap %5.1f  dur %6.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n))
        prev = e
    print()
PY
cat $D/table.txt; head -24 $D/one_update.txt
find $D -name "*kernel_trace.csv" -delete || true
