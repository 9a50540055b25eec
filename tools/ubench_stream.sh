#!/bin/bash
# Streaming-ceiling sweep on a GPU box: gpurun -- 'bash tools/ubench_stream.sh'.  Full sweep un-profiled, then the best
# configuration of every op once more under rocprofv3 (kernel trace; FETCH_SIZE and WRITE_SIZE in separate passes).
O=$GRAFT_REPO_ROOT/gpurun_out/r5_stream; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/ubench_stream
timeout 600 $B > $O/sweep.txt 2>&1
sed -n '/^# best per op and width/,/^$/p' $O/sweep.txt > $O/best.txt
cat $O/best.txt
# op W U nt k mode of the best line of every op (16-byte accesses) + the 12-byte copy
python3 - $O/best.txt > $O/sel.txt <<'PY'
import re, sys
ops = {"read": 0, "write": 1, "copy": 2, "triad": 3, "planes": 4}
modes = {"stride": 0, "contig": 1, "xcd": 2, "oneshot": 3, "oneshot_xcd": 4}
for l in open(sys.argv[1]):
    m = re.match(r"(\w+)\s+W=\s*(\d+) U=(\d) (\w+)\s+k=(\d)\(occ \d+\) (\w+)", l)
    if m and (m.group(2) in ("16", "12")):
        print(ops[m.group(1)], m.group(2), m.group(3), 1 if m.group(4) == "nt" else 0, m.group(5), modes[m.group(6)])
PY
i=0
while read sel; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$i -o t -- $B only $sel 20 > $O/t$i.log 2>&1
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f$i -o t -- $B only $sel 20 > $O/f$i.log 2>&1
  timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w$i -o t -- $B only $sel 20 > $O/w$i.log 2>&1
  echo "== $sel"; tail -1 $O/t$i.log
  python3 - $O/t$i $O/f$i $O/w$i <<'PY'
import csv, glob, sys
def rows(d, pat):
    out = []
    for f in glob.glob(d + "/**/*" + pat, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
st = [r for r in rows(sys.argv[1], "kernel_stats.csv") if "k_" in r.get("Name", "")]
for r in st: print("  rocprof", r["Name"][:60], "calls", r["Calls"], "avg_ns", r["AverageNs"])
for d, c in ((sys.argv[2], "FETCH_SIZE"), (sys.argv[3], "WRITE_SIZE")):
    v = [float(r["Counter_Value"]) for r in rows(d, "counter_collection.csv") if r.get("Counter_Name") == c and "k_" in r.get("Kernel_Name", "")]
    if v: print("  %s per launch (KiB units x 1024): %.1f MB over %d launches" % (c, sum(v) / len(v) * 1024 / 1e6, len(v)))
PY
done < $O/sel.txt 2>&1 | tee $O/verify.txt
find $O -name "*.csv" -size +2M -delete
