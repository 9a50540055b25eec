#!/bin/bash
# Reproduces the files under profiles/ on a GPU box (run from the repo root, e.g. `gpurun -- 'bash tools/refresh_profiles.sh'`):
# the bench lines of every mode, the rocprofv3 kernel trace of the default (Laplace) bench and the two PMC passes for the
# HBM traffic (FETCH_SIZE and WRITE_SIZE in separate runs, no other trace domains).  Outputs land in gpurun_out/profiles/;
# tools/rocpd_stats.py folds the rocpd databases into the text summaries that are committed.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles
mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py > $OUT/bench_laplace.json 2> $OUT/bench_laplace.err
timeout 300 python bench.py --mode riesz > $OUT/bench_riesz.json 2>/dev/null
timeout 300 python bench.py --mode color > $OUT/bench_color.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --frames-per-call 1 > $OUT/bench_laplace_perframe.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --streams 8 > $OUT/bench_laplace_8streams.json 2>/dev/null
timeout 400 python bench.py --no-cpu-baseline --mode riesz --width 3840 --height 2160 --levels 8 --steps 96 --warmup 32 > $OUT/bench_riesz_4k.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --profile-steps 0 --steps 128 --warmup 64"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o lap -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof -o lap_fetch -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof -o lap_write -- $B > /dev/null 2>&1
cd $ROOT
python tools/rocpd_stats.py $OUT/prof/lap_results.db | head -20
