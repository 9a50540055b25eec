mkdir -p gpurun_out/r8
B="timeout 300 python bench.py --no-cpu-baseline"
$B --mode color > gpurun_out/r8/col_default.json 2> gpurun_out/r8/col_default.err
$B --mode color --frames-per-call 1 > gpurun_out/r8/col_perframe.json 2>/dev/null
LVM_COL_THIN_DFT=0 $B --mode color --frames-per-call 1 > gpurun_out/r8/col_perframe_nothin.json 2>/dev/null
for f in gpurun_out/r8/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(d['value'], d['frame_roofline_frac'], d['ms_per_step'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e: print('ERR',e)
PY
done
