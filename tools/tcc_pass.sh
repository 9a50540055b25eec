#!/bin/bash
# L2 / fabric-side (TCC_EA) counters of the big kernels against the SAME counters of the streaming microbenchmark at its ceiling:
# is a kernel waiting for HBM?  Credit stalls (requests held back because the memory side has no credit = saturation), read-request
# occupancy (LEVEL / RDREQ = average latency in TCC cycles) and L2 hit rate, per kernel.   gpurun -- 'bash tools/tcc_pass.sh'
O=$GRAFT_REPO_ROOT/gpurun_out/r6_tcc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
A="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
B="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum"
run() {   # name, command...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $O/${n}_a -o t -- "$@" > $O/${n}_a.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $O/${n}_b -o t -- "$@" > $O/${n}_b.log 2>&1
}
S=$GRAFT_REPO_ROOT/tools/ubench_stream
run stream_copy $S only 2 16 1 1 8 3 20
run stream_read $S only 0 16 4 1 8 3 20
run stream_planes $S only 4 16 1 1 8 4 20
BN="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-subrecords --no-clock-probe --profile-steps 0 --steps 6 --warmup 2"
run laplace $BN --mode laplace
run riesz $BN --mode riesz
run color $BN --mode color
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
def fold(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); name = {}
        for r in csv.DictReader(open(f)):
            acc[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); name[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (did, c), v in acc.items():
            per[name[did].split("(")[0].replace("void lvm::", "").replace("void ", "")[:44]][c].append(v)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0].replace("void lvm::", "").replace("void ", "")[:44]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return per, dur
print("%-46s %8s %9s %9s %8s %12s %12s %10s" % ("kernel (largest launches)", "us", "RD GB/s", "WR GB/s", "L2 hit", "rd credit", "wr stall", "rd latency"))
print("%-46s %8s %9s %9s %8s %12s %12s %10s" % ("", "", "(x2 cal.)", "", "", "stall/req", "/req", "TCC cycles"))
for n in ("stream_read", "stream_copy", "stream_planes", "laplace", "riesz", "color"):
    pa, da = fold("%s/%s_a" % (O, n)); pb, db = fold("%s/%s_b" % (O, n))
    print("== " + n)
    rows = []
    for k in pa:
        if not (k.startswith("k_")) or k not in pb: continue
        top = lambda v: sorted(v)[-max(1, len(v) // 4):]          # the steady-state (largest) launches
        m = lambda d, c: sum(top(d[k][c])) / len(top(d[k][c])) if d[k].get(c) else 0.0
        us = sum(top(da[k])) / len(top(da[k])) if da.get(k) else 0.0
        if us < 40: continue
        rd, wr = m(pa, "TCC_EA0_RDREQ_sum"), m(pb, "TCC_EA0_WRREQ_sum")
        hit, miss = m(pa, "TCC_HIT_sum"), m(pa, "TCC_MISS_sum")
        rows.append((us, "%-46s %8.1f %9.0f %9.0f %8.3f %12.3f %12.3f %10.0f" % (k, us, rd * 128 / us / 1e3, wr * 64 / us / 1e3, hit / max(hit + miss, 1),
                     m(pa, "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum") / max(rd, 1), m(pb, "TCC_EA0_WRREQ_STALL_sum") / max(wr, 1), m(pb, "TCC_EA0_RDREQ_LEVEL_sum") / max(rd, 1))))
    for _, l in sorted(rows, reverse=True): print(l)
PY
