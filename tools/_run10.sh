mkdir -p gpurun_out/r11
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/r11/bench_laplace.json 2> gpurun_out/r11/bench_laplace.err
timeout 300 python bench.py --mode riesz > gpurun_out/r11/bench_riesz.json 2>/dev/null
timeout 300 python bench.py --mode color > gpurun_out/r11/bench_color.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --frames-per-call 1 > gpurun_out/r11/bench_laplace_perframe.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --streams 8 > gpurun_out/r11/bench_laplace_8streams.json 2>/dev/null
timeout 400 python bench.py --no-cpu-baseline --mode riesz --width 3840 --height 2160 --levels 8 --steps 96 --warmup 32 > gpurun_out/r11/bench_riesz_4k.json 2> gpurun_out/r11/bench_riesz_4k.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r11/prof -o lap -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --profile-steps 0 --steps 128 --warmup 64 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r11/prof -o lap_fetch -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --profile-steps 0 --steps 128 --warmup 64 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r11/prof -o lap_write -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --profile-steps 0 --steps 128 --warmup 64 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/r11/prof
for f in gpurun_out/r11/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(d['value'], d['frame_roofline_frac'], d['ms_per_step'], d.get('roofline'), d.get('cpu_baseline'))
except Exception as e: print('ERR',e)
PY
done
