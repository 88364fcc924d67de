"""The NODE fits in a rocprofv3 kernel trace of bench.py (Unicycle): span of every fit — its first obs -> state launch
to the end of its Adam step — and the kernel timeline of the shortest one.
    python tools/fit_table.py <run_kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:50]) for r in rows)
# the fit's weight gradients are the one launch of mlp_dw16 (or, NLBAC_MLP_DW16=0, of mlp_bwd_wide128) per RK step
idx = [i for i, k in enumerate(ks) if k[2].startswith(("mlp_dw16", "mlp_bwd_wide128"))]
spans = []
for i in idx:
    j = i
    while not ks[j][2].startswith("unicycle_state"):
        j -= 1
    j -= 1
    e = i
    while not ks[e][2].startswith("adam_fused"):
        e += 1
    spans.append((ks[e][1] - ks[j][0], j, e))
spans.sort()
print("NODE fits in the trace: %d; span (first state launch -> end of the fit's Adam step): min %.0f us, median %.0f us, "
      "max %.0f us (the largest are warm-up fits: allocation, first launches)"
      % (len(spans), spans[0][0] / 1e3, spans[len(spans) // 2][0] / 1e3, spans[-1][0] / 1e3))
_, j, e = spans[0]
t0 = ks[j][0]
print("the shortest one:   start (us)   duration")
for s, en, name in ks[j:e + 1]:
    print("  %10.1f  %8.1f us  %s" % ((s - t0) / 1e3, (en - s) / 1e3, name))
