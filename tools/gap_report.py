"""GPU idle time between consecutive kernels of a rocprofv3 kernel trace (``*_kernel_trace.csv``), per update
(updates are delimited by the minibatch draw, ``sample_rows``; a NODE fit draws a second time, so pick a window
between two fits — tools/update_table.py prints the kernel table of the same window).  Usage: python tools/gap_report.py TRACE.csv [first last]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44])
                for r in rows)
    marks = [i for i, k in enumerate(ks) if k[2].startswith("sample_rows")]
    a = int(sys.argv[2]) if len(sys.argv) > 2 else 61
    b = int(sys.argv[3]) if len(sys.argv) > 3 else 69
    seg = ks[marks[a]:marks[b]]
    n = b - a
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    print("updates %d..%d: span %.1f us/update, kernels busy %.1f us, idle %.1f us, %.1f launches/update"
          % (a, b, span / 1e3 / n, busy / 1e3 / n, (span - busy) / 1e3 / n, len(seg) / n))
    gaps, cnt = collections.Counter(), collections.Counter()
    for (s0, e0, n0), (s1, e1, n1) in zip(seg, seg[1:]):
        if s1 > e0:
            gaps[(n0, n1)] += s1 - e0
            cnt[(n0, n1)] += 1
    for k, v in gaps.most_common(16):
        print("  %-44s -> %-44s %6.1f us/update  (%.2f x %.1f us)" % (k[0], k[1], v / 1e3 / n, cnt[k] / n, v / 1e3 / cnt[k]))


if __name__ == "__main__":
    main()
