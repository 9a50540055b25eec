#!/usr/bin/env python3
"""Joins tools/hbm_counter_calib's known byte counts with the FETCH_SIZE / WRITE_SIZE passes of tools/calib.sh.
Usage: hbm_counter_calib.py known.jsonl DIR_FETCH DIR_WRITE > profiles/r03_hbm_counter_calibration.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read(d, counter):
    per = defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0])] += float(r["Counter_Value"])
    out = defaultdict(list)
    for (_, k), v in per.items():
        out[k].append(v)
    return out


def main():
    known = [json.loads(ln) for ln in open(sys.argv[1]) if ln.startswith("{")]
    fetch, write = read(sys.argv[2], "FETCH_SIZE"), read(sys.argv[3], "WRITE_SIZE")
    rows = []
    for k in known:
        fe = sorted(v for n, vs in fetch.items() if n.endswith(k["kernel"]) for v in vs)
        wr = sorted(v for n, vs in write.items() if n.endswith(k["kernel"]) for v in vs)
        med = lambda xs: xs[len(xs) // 2] if xs else None  # noqa: E731
        f_kb, w_kb = med(fe), med(wr)
        row = dict(k, FETCH_SIZE_KB=f_kb, WRITE_SIZE_KB=w_kb)
        if k["read_bytes"] and f_kb:
            row["fetch_factor"] = round(k["read_bytes"] / (f_kb * 1024.0), 4)      # true bytes / counter bytes
        if k["write_bytes"] and w_kb:
            row["write_factor"] = round(k["write_bytes"] / (w_kb * 1024.0), 4)
        rows.append(row)
    wide = [r["fetch_factor"] for r in rows if r.get("fetch_factor") and r["bytes_per_lane"] >= 8 and "cached" not in r["kernel"]]
    narrow = [r["fetch_factor"] for r in rows if r.get("fetch_factor") and r["bytes_per_lane"] == 4 and "cached" not in r["kernel"]]
    wf = [r["write_factor"] for r in rows if r.get("write_factor") and "cached" not in r["kernel"]]
    print(json.dumps({"note": "true bytes / (counter KB x 1024), rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, MI355X; "
                              "1 GiB footprints (4 x the Infinity Cache) unless the kernel name ends in _cached (64 MiB re-read by six launches)",
                      "factors": {"fetch_4B_per_lane": narrow and round(sum(narrow) / len(narrow), 4), "fetch_8B_and_wider": wide and round(sum(wide) / len(wide), 4),
                                  "write": wf and round(sum(wf) / len(wf), 4)},
                      "kernels": rows}, indent=1))


if __name__ == "__main__":
    main()
