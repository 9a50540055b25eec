mkdir -p gpurun_out/r22
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r22/gpu_tests.txt 2>&1
tail -3 gpurun_out/r22/gpu_tests.txt
B="timeout 300 python bench.py --no-cpu-baseline"
$B --mode riesz > gpurun_out/r22/rz_default.json 2> gpurun_out/r22/rz_default.err
LVM_RZ_SPLIT2=0 $B --mode riesz > gpurun_out/r22/rz_nosplit2.json 2>/dev/null
for f in gpurun_out/r22/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(d['value'], d['frame_roofline_frac'], d['ms_per_step'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e: print('ERR',e)
PY
done
