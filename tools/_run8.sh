mkdir -p gpurun_out/r9
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r9/gpu_tests.txt 2>&1
B="timeout 300 python bench.py --no-cpu-baseline"
$B --mode color > gpurun_out/r9/col_default.json 2> gpurun_out/r9/col_default.err
LVM_COL_OUT_ROWS=0 $B --mode color > gpurun_out/r9/col_tiled.json 2>/dev/null
LVM_COL_OUT_ROWS=4 $B --mode color > gpurun_out/r9/col_rows4.json 2>/dev/null
LVM_COL_OUT_ROWS=16 $B --mode color > gpurun_out/r9/col_rows16.json 2>/dev/null
$B --mode color --frames-per-call 1 > gpurun_out/r9/col_perframe.json 2>/dev/null
tail -3 gpurun_out/r9/gpu_tests.txt
for f in gpurun_out/r9/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(d['value'], d['frame_roofline_frac'], d['ms_per_step'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e: print('ERR',e)
PY
done
