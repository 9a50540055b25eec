#!/usr/bin/env python3
"""Per-kernel-family busy time and pairwise overlap from a rocprofv3 --kernel-trace CSV (start / end timestamps of every
dispatch): how much of the table-conversion kernel's run time coincided with the HBM-bound kernels when the two were issued
on different streams (LVM_LAP_CHUNKS).  Usage: overlap_report.py DIR"""
import csv
import glob
import os
import sys


def fam(name):
    n = name.split("(")[0]
    if "k_down0_lut_rows" in n or "k_lab_planes" in n:
        return "table"
    return "rest"


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def inter(A, B):
    i = j = 0
    tot = 0
    while i < len(A) and j < len(B):
        lo, hi = max(A[i][0], B[j][0]), min(A[i][1], B[j][1])
        if hi > lo:
            tot += hi - lo
        if A[i][1] < B[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows = [r for r in rows if "lvm::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 3:]                      # steady state: the last two thirds of the run
    iv = {"table": [], "rest": []}
    queues = {"table": set(), "rest": set()}
    for r in rows:
        k = fam(r["Kernel_Name"])
        iv[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        queues[k].add(r.get("Queue_Id", "?"))
    U = {k: union(v) for k, v in iv.items()}
    busy = {k: sum(b - a for a, b in v) for k, v in U.items()}
    span = max(b for v in U.values() for a, b in v) - min(a for v in U.values() for a, b in v)
    both = inter(U["table"], U["rest"])
    print("dispatches: table %d (queues %s), rest %d (queues %s)" % (len(iv["table"]), sorted(queues["table"]), len(iv["rest"]), sorted(queues["rest"])))
    print("span %.3f ms; table kernels busy %.3f ms, other kernels busy %.3f ms, both at once %.3f ms (%.1f %% of the table time), neither %.3f ms"
          % (span / 1e6, busy["table"] / 1e6, busy["rest"] / 1e6, both / 1e6, 100.0 * both / max(busy["table"], 1), (span - busy["table"] - busy["rest"] + both) / 1e6))


if __name__ == "__main__":
    main()
