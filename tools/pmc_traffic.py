#!/usr/bin/env python3
"""HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc.sh -> profiles/rNN_pmc_traffic_<mode>.json
(the file bench.py reads for `roofline.traffic`).  Usage: pmc_traffic.py MODE KEY DIR_P0 DIR_FETCH DIR_WRITE > out.json
Counter values are KB summed over the TCC instances; per-launch averages over the launches of the steady-state grid
(the largest launches of each symbol)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

NAMES = {  # kernel symbol prefix -> bench.py report name (default flavour: FL_LUT_FAST = 0)
    "k_lap_final_v4<true, false, 0>": "lap_final", "k_down0_lut_rows<0>": "lap_down0_lut", "k_down0_rows<true, 0>": "lap_down0", "k_down0_rows<false, 1>": "col_down0",
    "k_lab_planes": "lab_lut", "k_lap_up<false, 1>": "lap_up_l1", "k_lap_iir_levels": "lap_iir", "k_lap_collapse": "lap_collapse", "k_pyr_down_rows": "pyr_down_rows_l1",
    "k_rz_final<true, 0, true, true, false>": "rz_final", "k_rz_collapse_strips<true, 0, false>": "rz_final", "k_rz_collapse_strips<false, 0, false>": "rz_collapse_l1", "k_rz_blur_amp4<false>": "rz_blur_amp", "k_rz_blur_strips": "rz_blur_amp", "k_rz_phase4<false>": "rz_phase", "k_rz_phase<false>": "rz_phase_small",
    "k_rz_split_rows": "rz_split_l0", "k_col_out_rows<true, false>": "col_out", "k_col_out_rows<false, false>": "col_minmax",
    "k_down01_rows": "col_down01", "k_col_out_strips<true, false>": "col_out_u2", "k_col_out_strips<false, false>": "col_minmax_u2",
}
HERE = os.path.dirname(os.path.abspath(__file__))


def calibration():
    """true bytes / counter bytes, measured by tools/hbm_counter_calib (tools/calib.sh) on known byte counts"""
    try:
        f = json.load(open(os.path.join(HERE, "..", "profiles", "r03_hbm_counter_calibration.json")))["factors"]
        return float(f["fetch_8B_and_wider"]), float(f["write"])
    except Exception:
        return 2.0, 1.0


def short(name):
    name = name.split("(")[0]
    for pre in ("void lvm::", "lvm::"):
        if name.startswith(pre):
            name = name[len(pre):]
    return name


def read(d, counter):
    per = defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[(r["Dispatch_Id"], short(r["Kernel_Name"]))] += float(r["Counter_Value"])
    out = defaultdict(list)
    for (_, k), v in per.items():
        out[k].append(v)
    return out


def main():
    mode, key, p0, pf, pw = sys.argv[1:6]
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(p0, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    fetch, write = read(pf, "FETCH_SIZE"), read(pw, "WRITE_SIZE")
    ff, wf = calibration()
    kernels = {}
    for sym in sorted(dur, key=lambda k: -sum(dur[k])):
        name = next((v for k, v in NAMES.items() if sym.startswith(k)), None)
        if name is None or name in kernels:      # (symbols are visited by total time: the steady-state variant comes first)
            continue
        def big(xs):      # steady-state launches only: the large grid, without isolated cold-start spikes
            if not xs:
                return []
            srt = sorted(xs, reverse=True)
            ref = srt[0]
            for v in srt:     # the largest value that at least three launches come close to (an isolated spike is not the reference)
                if sum(1 for x in srt if x >= 0.8 * v) >= min(3, len(srt)):
                    ref = v
                    break
            return [x for x in xs if 0.5 * ref <= x <= 1.3 * ref]
        d, fe, wr = big(dur[sym]), big(fetch.get(sym, [])), big(write.get(sym, []))
        if not d:
            continue
        fk, wk = (sum(fe) / len(fe) if fe else 0.0), (sum(wr) / len(wr) if wr else 0.0)
        kernels[name] = {"symbol": sym, "launches": len(d), "rocprof_avg_us": round(sum(d) / len(d), 2), "FETCH_SIZE_KB": round(fk, 1),
                         "WRITE_SIZE_KB": round(wk, 1), "hbm_bytes_per_launch": int((ff * fk + wf * wk) * 1024)}
    print(json.dumps({"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no other trace domain) of "
                              "bench.py on MI355X; KB summed over the TCC instances, averaged over the steady-state launches.  "
                              "hbm_bytes_per_launch = (%.3f x FETCH_SIZE + %.3f x WRITE_SIZE) x 1024: the factors are true bytes / counter bytes "
                              "measured on known byte counts at 4, 8, 12, 16 and 8 + 16 bytes per lane, 1 GiB and 64 MiB footprints "
                              "(profiles/r03_hbm_counter_calibration.json): FETCH_SIZE reports exactly half of what is read at every width, "
                              "Infinity-Cache hits included; WRITE_SIZE is exact.  (The round-2 files applied no factor: their read "
                              "halves are 2 x too small.)" % (ff, wf),
                      "key": key, "mode": mode, "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
