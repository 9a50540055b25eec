mkdir -p gpurun_out/r3
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3/gpu_tests.txt 2>&1
B="timeout 300 python bench.py --no-cpu-baseline"
$B > gpurun_out/r3/lap_default.json 2> gpurun_out/r3/lap_default.err
LVM_UP_DEPTH=1 $B > gpurun_out/r3/lap_up1.json 2>/dev/null
LVM_UP_DEPTH=4 $B > gpurun_out/r3/lap_up4.json 2>/dev/null
LVM_FIN_ROWS=8 $B > gpurun_out/r3/lap_fin8.json 2>/dev/null
LVM_FIN_ROWS=2 $B > gpurun_out/r3/lap_fin2.json 2>/dev/null
$B --frames-per-call 1 > gpurun_out/r3/lap_perframe.json 2>/dev/null
LVM_FIN_ROWS=8 $B --frames-per-call 1 > gpurun_out/r3/lap_perframe_fin8.json 2>/dev/null
$B --frames-per-call 32 --ring 64 > gpurun_out/r3/lap_T32.json 2>/dev/null
$B --frames-per-call 8 > gpurun_out/r3/lap_T8.json 2>/dev/null
$B --streams 8 > gpurun_out/r3/lap_8s.json 2>/dev/null
tail -3 gpurun_out/r3/gpu_tests.txt
for f in gpurun_out/r3/lap_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(d['value'], d['frame_roofline_frac'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e: print('ERR',e)
PY
done
