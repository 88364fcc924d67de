"""Per-update kernel table from a rocprofv3 kernel trace of bench.py (updates are delimited by the minibatch draw,
sample_rows_kernel):  python tools/update_table.py <run_kernel_trace.csv> [first last]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]) for r in rows)
marks = [i for i, k in enumerate(ks) if k[2].startswith("sample_rows")]
a, b = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (61, 69)
seg = ks[marks[a]:marks[b]]
n = b - a
c, t = collections.Counter(), collections.Counter()
for s, e, name in seg:
    c[name] += 1
    t[name] += e - s
for name, v in sorted(t.items(), key=lambda kv: -kv[1]):
    print("%-46s %5.2f /update  %7.1f us/update  avg %.1f us" % (name, c[name] / n, v / n / 1e3, v / c[name] / 1e3))
mf = ("mlp_", "node_rk", "node_rr", "concat_rk", "concat_rr", "node_adj")
print("updates %d..%d: span %.1f us/update, busy %.1f us/update, %.1f launches/update, of which not MFMA tile kernels: %.1f (%.1f us)"
      % (a, b, (seg[-1][1] - seg[0][0]) / n / 1e3, sum(t.values()) / n / 1e3, len(seg) / n,
         sum(v for k, v in c.items() if not k.startswith(mf)) / n, sum(v for k, v in t.items() if not k.startswith(mf)) / n / 1e3))
