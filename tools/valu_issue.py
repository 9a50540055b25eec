#!/usr/bin/env python3
"""Vector-issue share of every kernel from the SQ counter summaries of tools/refresh_profiles.sh (profiles/rNN_rocprof_<mode>_kernels_and_sq_counters.txt):
SQ_INSTS_VALU wave-instructions per launch x issue cycles per instruction / (1024 SIMDs x clock) against the kernel's rocprofv3 duration.
Issue cost: 2.95 cycles for v_fma / v_add / v_mul, 4.1-4.5 for conversions, integer, DPP and select forms, 4.96 for packed FP32
(profiles/r02_ubench_valu_issue_rates.txt) -- 3.0 is used as the floor and 4.2 as the typical mix, at the 2.4 GHz maximum clock (a lower real
clock only raises the share).  Usage: valu_issue.py SUMMARY.txt [...] > profiles/rNN_valu_issue.json"""
import json
import re
import sys

SIMDS, GHZ = 1024, 2.4
args = sys.argv[1:]
clock_src = "assumed: the part's maximum"
if len(args) >= 2 and args[0] == "--clock-mhz":          # round 6: the clock bench.py measured over its timed region (lvm_debug_clock_probe_*)
    GHZ = float(args[1]) / 1000.0
    clock_src = "measured by bench.py over its timed region: s_memtime / s_memrealtime (clock_mhz of the same refresh run)"
    args = args[2:]
out = {}
for path in args:
    lines = open(path).read().splitlines()
    hdr = lines[0]
    # fixed-width columns: find the column starts of the numeric fields from the header
    names = hdr.split()
    for ln in lines[1:]:
        m = re.match(r"(.{44})\s+(\d+)\s+([\d.]+)\s+(.*)$", ln)
        if not m:
            continue
        kern, calls, avg_us, rest = m.group(1).strip(), int(m.group(2)), float(m.group(3)), m.group(4).split()
        cols = dict(zip(names[3:], rest))
        valu = next((float(v) for k, v in cols.items() if k.endswith("SQ_INSTS_VALU")), None)
        if valu is None or not kern.startswith("k_") or kern.startswith("k_clock_probe") or calls < 4:
            continue
        lo = valu * 3.0 / SIMDS / (GHZ * 1e3)
        ty = valu * 4.2 / SIMDS / (GHZ * 1e3)
        out[kern] = {"rocprof_avg_us": avg_us, "valu_wave_instructions_per_launch": valu, "valu_issue_us_at_3.0_cycles": round(lo, 1),
                     "valu_issue_us_at_4.2_cycles": round(ty, 1), "valu_issue_share_floor": round(lo / avg_us, 3), "valu_issue_share_typical": round(min(ty / avg_us, 1.0), 3)}
print(json.dumps({"note": __doc__.strip().split("\n\n")[0], "simds": SIMDS, "clock_ghz": GHZ, "clock_source": clock_src, "kernels": out}, indent=1))
