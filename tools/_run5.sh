mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r6/gpu_tests.txt 2>&1
B="timeout 300 python bench.py --no-cpu-baseline"
$B > gpurun_out/r6/lap_default.json 2> gpurun_out/r6/lap_default.err
LVM_UP_ROWS_MAX_BLOCKS=100000000 $B > gpurun_out/r6/lap_rowsall.json 2>/dev/null
$B --streams 8 > gpurun_out/r6/lap_8s.json 2>/dev/null
$B --frames-per-call 1 > gpurun_out/r6/lap_perframe.json 2>/dev/null
tail -3 gpurun_out/r6/gpu_tests.txt
for f in gpurun_out/r6/lap_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(d['value'], d['frame_roofline_frac'], d['ms_per_step'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e: print('ERR',e)
PY
done
