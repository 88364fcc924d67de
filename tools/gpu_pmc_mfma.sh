#!/bin/bash
# On the GPU box: MFMA-busy cycles per kernel (one SQ counter pass, kernel trace only).
#   bash tools/gpu_pmc_mfma.sh <tag> [bench args]
set -e
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; shift
export TMPDIR=/tmp
D=gpurun_out/pmc_${TAG}_mfma
mkdir -p $D
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D -o run -- python3 bench.py --no-cpu-baseline --profile-steps 0 "$@" > $D/bench.json 2> $D/bench.err || { tail -20 $D/bench.err; exit 1; }
python3 - "$D" <<'PY'
import csv, glob, os, re, sys, json
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[name] += 1
out = {}
for k, v in acc.items():
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
        continue
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    # SQ_VALU_MFMA_BUSY_CYCLES sums over SIMDs (4 per CU x 256 CUs); GRBM_GUI_ACTIVE is per-dispatch wall cycles
    out[k] = dict(launches=cnt[k], mfma_busy_cycles=v["SQ_VALU_MFMA_BUSY_CYCLES"], gui_active_cycles=gui,
                  mfma_util=v["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024.0) if gui else None)
json.dump(dict(note="mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs), summed over all launches of the kernel",
               kernels=out), open(os.path.join(os.path.dirname(d), os.path.basename(d) + ".json"), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"]):
    print("%-48s n=%4d  mfma_util %.3f" % (k[:48], v["launches"], v["mfma_util"] or 0))
PY
find $D -name "*.csv" -size +8M -delete || true
