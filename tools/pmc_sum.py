#!/usr/bin/env python3
"""Per-kernel sums of one counter from a `rocprofv3 --pmc <COUNTER> --output-format csv` run.

  python tools/pmc_sum.py <dir with *_counter_collection.csv> <COUNTER> > out.csv

Output: kernel, dispatches, sum, max per dispatch (the counter's own unit; FETCH_SIZE
and WRITE_SIZE are in KB)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root, counter = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no *_counter_collection.csv under " + root)
    per_dispatch = defaultdict(float)          # (kernel, dispatch id) -> value
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                per_dispatch[(row["Kernel_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for (kernel, _), v in per_dispatch.items():
        a = agg[kernel]
        a[0] += 1
        a[1] += v
        a[2] = max(a[2], v)
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "dispatches", "sum_%s" % counter, "max_per_dispatch"])
    for kernel, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([kernel, a[0], "%.1f" % a[1], "%.1f" % a[2]])


if __name__ == "__main__":
    main()
