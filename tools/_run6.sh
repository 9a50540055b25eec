mkdir -p gpurun_out/r7
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r7/gpu_tests.txt 2>&1
B="timeout 300 python bench.py --no-cpu-baseline"
$B --mode color > gpurun_out/r7/col_default.json 2> gpurun_out/r7/col_default.err
LVM_COL_THIN_DFT=0 $B --mode color > gpurun_out/r7/col_nothin.json 2>/dev/null
$B --mode riesz > gpurun_out/r7/rz_default.json 2>/dev/null
$B --mode color --frames-per-call 1 > gpurun_out/r7/col_perframe.json 2>/dev/null
$B --mode riesz --frames-per-call 1 > gpurun_out/r7/rz_perframe.json 2>/dev/null
tail -3 gpurun_out/r7/gpu_tests.txt
for f in gpurun_out/r7/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(d['value'], d['frame_roofline_frac'], d['ms_per_step'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e: print('ERR',e)
PY
done
