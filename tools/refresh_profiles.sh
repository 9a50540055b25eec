#!/bin/bash
# Reproduces the round-6 files under profiles/ on a GPU box (`gpurun -- 'bash tools/refresh_profiles.sh'`): the bench
# lines of every mode (driver-shaped and default), the rocprofv3 kernel-trace summaries and the PMC passes (FETCH_SIZE,
# WRITE_SIZE, SQ counters: each in its own run, never combined with other trace domains).  Outputs land in gpurun_out/profiles_r05/; copy what is to be judged
# into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=r06
O=$ROOT/gpurun_out/profiles_$R; mkdir -p $O; cd $ROOT
# (round 6: a step is one 32-frame call; `--steps 20 --warmup 5` IS the default and the driver's command)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${R}_bench_laplace_driver_shape.json 2> $O/err.txt
timeout 600 python bench.py --steps 60 --warmup 10 --no-subrecords > $O/${R}_bench_laplace.json 2>> $O/err.txt
timeout 400 python bench.py --mode riesz --no-subrecords > $O/${R}_bench_riesz.json 2>> $O/err.txt
timeout 400 python bench.py --mode color --no-subrecords > $O/${R}_bench_color.json 2>> $O/err.txt
tools/ubench_valu > $O/${R}_ubench_valu_issue_rates.txt 2>&1
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ2="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
for m in ${MODES:-laplace riesz color}; do
  bash tools/pmc.sh profiles_$R/pmc_$m "--mode $m --steps 6 --warmup 2" "$SQ1" "$SQ2" "FETCH_SIZE" "WRITE_SIZE" > /dev/null 2>&1
  cp $O/pmc_$m/summary.txt $O/${R}_rocprof_${m}_kernels_and_sq_counters.txt
  cp $O/pmc_$m/p0/t_kernel_stats.csv $O/${R}_rocprof_${m}_kernel_stats.csv
  python tools/pmc_traffic.py $m "$m|1920x1080|L6|B1|T32" $O/pmc_$m/p0 $O/pmc_$m/p3 $O/pmc_$m/p4 > $O/${R}_pmc_traffic_$m.json
  rm -rf $O/pmc_$m/p1 $O/pmc_$m/p2 $O/pmc_$m/p3 $O/pmc_$m/p4       # the per-dispatch counter CSVs are large; the summaries stay
done
# BASELINE configs[4] (Riesz 3840x2160, 8 levels, 16 frames per call): kernel stats + calibrated traffic for the cfg4 sub-record's roofline
bash tools/pmc.sh profiles_$R/pmc_riesz_4k "--mode riesz --width 3840 --height 2160 --levels 8 --frames-per-call 16 --ring 16 --steps 4 --warmup 1" "FETCH_SIZE" "WRITE_SIZE" > /dev/null 2>&1
cp $O/pmc_riesz_4k/p0/t_kernel_stats.csv $O/${R}_rocprof_riesz_4k_kernel_stats.csv
python tools/pmc_traffic.py riesz "riesz|3840x2160|L8|B1|T16" $O/pmc_riesz_4k/p0 $O/pmc_riesz_4k/p1 $O/pmc_riesz_4k/p2 > $O/${R}_pmc_traffic_riesz_3840x2160.json
rm -rf $O/pmc_riesz_4k/p1 $O/pmc_riesz_4k/p2
CLK=$(python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/${R}_bench_laplace.json") if l.startswith("{")][0]); print(d.get("clock_mhz") or 2400)
except Exception:
    print(2400)
PY
)
python tools/valu_issue.py --clock-mhz $CLK $O/${R}_rocprof_laplace_kernels_and_sq_counters.txt $O/${R}_rocprof_riesz_kernels_and_sq_counters.txt $O/${R}_rocprof_color_kernels_and_sq_counters.txt > $O/${R}_valu_issue.json
# L2 / fabric-side counters of the big kernels against the streaming microbenchmark at its ceiling (is a kernel waiting for HBM?)
[ -x tools/ubench_stream ] || hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o tools/ubench_stream
bash tools/tcc_pass.sh > $O/${R}_tcc_ea_counters.txt 2>&1
rm -rf $ROOT/gpurun_out/r6_tcc
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k variant_envelope 2>&1 | grep "HIP vs oracle" > $O/${R}_variant_envelope_gpu.txt
timeout 300 python tools/tiling_bench.py 2>&1 | grep -v amdgpu.ids > $O/${R}_tiling_one_gpu.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/${R}_pytest_gpu.txt
ls -la $O | head -40
