mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gpu_tests.txt 2>&1
B="timeout 300 python bench.py --no-cpu-baseline"
$B > gpurun_out/r4/lap_default.json 2> gpurun_out/r4/lap_default.err
LVM_UP_ROWS=0 $B > gpurun_out/r4/lap_uptiled.json 2>/dev/null
LVM_UP_W4_MIN=0 $B > gpurun_out/r4/lap_w4all.json 2>/dev/null
LVM_UP_W4_MIN=100000000000 $B > gpurun_out/r4/lap_w2all.json 2>/dev/null
$B --frames-per-call 1 > gpurun_out/r4/lap_perframe.json 2>/dev/null
$B --frames-per-call 32 --ring 64 > gpurun_out/r4/lap_T32.json 2>/dev/null
$B --streams 8 > gpurun_out/r4/lap_8s.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4/prof -o lap -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --profile-steps 0 --steps 64 --warmup 32 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/r4/gpu_tests.txt
for f in gpurun_out/r4/lap_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(d['value'], d['frame_roofline_frac'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e: print('ERR',e)
PY
done
ls gpurun_out/r4/prof | head
