#!/usr/bin/env python3
"""Fold rocprofv3 CSV output (--kernel-trace [--pmc ...] --output-format csv) into per-kernel rows:
average duration from *_kernel_trace.csv, summed-over-dimensions counter values per dispatch averaged per kernel
from *_counter_collection.csv.  Usage: pmc_fold.py DIR [DIR ...]  (every DIR = one rocprofv3 pass)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for pre in ("void lvm::", "lvm::"):
        if name.startswith(pre):
            name = name[len(pre):]
    return name[:44]


def main():
    dur = defaultdict(list)
    cnt = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(float)
            for r in csv.DictReader(open(f)):
                per[(r["Dispatch_Id"], short(r["Kernel_Name"]), r["Counter_Name"])] += float(r["Counter_Value"])
            for (_, k, c), v in per.items():
                cnt[k][c].append(v)
    names = sorted(dur, key=lambda k: -sum(dur[k]))
    counters = sorted({c for k in cnt for c in cnt[k]})
    print("%-44s %6s %9s " % ("kernel", "calls", "avg_us") + " ".join("%14s" % c[-14:] for c in counters))
    for k in names:
        v = dur[k]
        # the first pass's durations only would be cleaner; all passes are averaged (they agree within a few %)
        print("%-44s %6d %9.2f " % (k, len(v), sum(v) / len(v)) +
              " ".join("%14.4g" % (sum(cnt[k][c]) / len(cnt[k][c])) if cnt[k].get(c) else "%14s" % "-" for c in counters))


if __name__ == "__main__":
    main()
