#!/bin/bash
# Colour-mode part of tools/refresh_profiles.sh (round 3, after the strip output kernels with both pyrUps inside and the two-level
# first pass): bench line, rocprofv3 kernel-trace summary, SQ counters, calibrated HBM traffic, and a 4K / 4-stream spot check.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=r03
O=$ROOT/gpurun_out/profiles_$R; mkdir -p $O; cd $ROOT
timeout 400 python bench.py --mode color --no-subrecords > $O/${R}_bench_color.json 2> $O/err_color.txt
timeout 300 python bench.py --mode color --width 3840 --height 2160 --levels 6 --no-subrecords --steps 64 --warmup 32 > $O/${R}_bench_color_4k.json 2>> $O/err_color.txt
timeout 300 python bench.py --mode color --streams 4 --no-subrecords --steps 64 --warmup 32 > $O/${R}_bench_color_4streams.json 2>> $O/err_color.txt
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ2="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
m=color
bash tools/pmc.sh profiles_$R/pmc_$m "--mode $m --steps 128 --warmup 32" "$SQ1" "$SQ2" "FETCH_SIZE" "WRITE_SIZE" > /dev/null 2>&1
cp $O/pmc_$m/summary.txt $O/${R}_rocprof_${m}_kernels_and_sq_counters.txt
cp $O/pmc_$m/p0/t_kernel_stats.csv $O/${R}_rocprof_${m}_kernel_stats.csv
python tools/pmc_traffic.py $m "$m|1920x1080|L6|B1|T32" $O/pmc_$m/p0 $O/pmc_$m/p3 $O/pmc_$m/p4 > $O/${R}_pmc_traffic_$m.json
rm -rf $O/pmc_$m/p1 $O/pmc_$m/p2 $O/pmc_$m/p3 $O/pmc_$m/p4
python - <<PY
import json
for f in ("${R}_bench_color.json", "${R}_bench_color_4k.json", "${R}_bench_color_4streams.json"):
    try:
        d = json.load(open("$O/" + f)); print(f, d["value"], d.get("verified"), d["roofline"]["kernel"], d["roofline"]["frac"])
    except Exception as e:
        print(f, "FAILED", e)
PY
cat $O/${R}_pmc_traffic_$m.json | head -40
