#!/bin/bash
# Kernel trace of the per-frame (T = 1) schedule: per-kernel device durations and the gaps between consecutive kernels of a frame.
#   gpurun -- 'bash tools/t1_trace.sh laplace'
M=${1:-laplace}
O=$GRAFT_REPO_ROOT/gpurun_out/r5_t1trace_$M; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/p0 -o t -- python $GRAFT_REPO_ROOT/bench.py --mode $M --frames-per-call 1 --ring 32 --steps 300 --warmup 32 --no-cpu-baseline --no-subrecords --profile-steps 0 --ramp-ms 0 > $O/p0.log 2>&1
python3 - $O/p0 <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "lvm::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                      # the timed region's half
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    n = a["Kernel_Name"].split("(")[0].replace("void lvm::", "")[:40]
    dur[n].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
    gap[n].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
tot_d = tot_g = 0
per_frame = {}
for n in dur:
    d = sum(dur[n]) / len(dur[n]); g = sorted(gap[n])[len(gap[n]) // 2]
    print("%-42s launches %5d  duration %7.2f us  gap to the next kernel (median) %6.2f us" % (n, len(dur[n]), d, g))
nf = max(len(v) for v in dur.values())
print("per frame: kernel time %.1f us, gaps %.1f us" % (sum(sum(v) for v in dur.values()) / nf, sum(sum(sorted(v)[:len(v)]) for v in gap.values()) / nf))
PY
