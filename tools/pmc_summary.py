"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; both in KiB).

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte read requests at 64 bytes,
so the read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Output: JSON
{kernel: {launches, fetch_bytes_per_launch, write_bytes_per_launch, hbm_bytes_per_launch}}.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(d, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under " + d
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
                name = re.sub(r"<.*", "", name)
                acc[name] += float(row["Counter_Value"])
                cnt[name] += 1
    return acc, cnt


def table(fd, wd):
    f_acc, f_cnt = per_kernel(fd, "FETCH_SIZE")
    w_acc, w_cnt = per_kernel(wd, "WRITE_SIZE")
    out = {}
    for k in sorted(f_acc, key=lambda k: -f_acc[k]):
        n = f_cnt[k]
        fetch = 2.0 * 1024.0 * f_acc[k] / n
        write = 1024.0 * w_acc.get(k, 0.0) / max(1, w_cnt.get(k, 0))
        out[k] = dict(launches=n, fetch_bytes_per_launch=fetch, write_bytes_per_launch=write,
                      hbm_bytes_per_launch=fetch + write)
    return out, sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out.values())


def main():
    """pmc_summary.py <fetch dir> <write dir> <updates of that process> [<fetch dir 2> <write dir 2> <updates 2>]
    With the second (shorter) process: per_update_bytes is the DIFFERENCE of the two processes' totals over the
    difference of their update counts (one-time traffic cancels); without it, the total over the update count."""
    fd, wd = sys.argv[1], sys.argv[2]
    n_updates = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # updates the profiled (lean) process ran
    out, total = table(fd, wd)
    res = dict(note="FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B) + WRITE_SIZE, KiB -> bytes, mean per launch",
               kernels=out)
    if len(sys.argv) > 6:
        out2, total2 = table(sys.argv[4], sys.argv[5])
        n2 = int(sys.argv[6])
        res.update(updates=n_updates - n2, per_update_bytes=(total - total2) / (n_updates - n2),
                   one_time_bytes=total - n_updates * (total - total2) / (n_updates - n2),
                   per_update_note="all kernels of two lean bench processes (bench.py --lean: exactly warm-up + timed updates, "
                                   "NODE fits included) that differ by %d updates: the difference of their totals over that "
                                   "count — what a process does once (zero fills at allocation, packing) cancels and is "
                                   "reported as one_time_bytes" % (n_updates - n2),
                   per_update_by_kernel={k: (v["hbm_bytes_per_launch"] * v["launches"]
                                             - out2.get(k, dict(hbm_bytes_per_launch=0, launches=0))["hbm_bytes_per_launch"]
                                             * out2.get(k, dict(launches=0))["launches"]) / (n_updates - n2) for k, v in out.items()})
    elif n_updates:
        res.update(updates=n_updates, per_update_bytes=total / n_updates,
                   per_update_note="all kernels of a lean bench process (bench.py --lean: exactly warm-up + timed updates, "
                                   "NODE fits included) divided by its update count (one-time traffic included)")
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
