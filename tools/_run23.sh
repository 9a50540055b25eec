mkdir -p gpurun_out/r24
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r24/gpu_tests.txt 2>&1
tail -3 gpurun_out/r24/gpu_tests.txt
B="timeout 300 python bench.py --no-cpu-baseline"
$B > gpurun_out/r24/lap_default.json 2> gpurun_out/r24/lap_default.err
$B --frames-per-call 16 --ring 32 > gpurun_out/r24/lap_T16.json 2>/dev/null
$B --mode color > gpurun_out/r24/col_default.json 2>/dev/null
for f in gpurun_out/r24/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(d['value'], d['frame_roofline_frac'], d['ms_per_step'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e: print('ERR',e)
PY
done
