#!/bin/bash
# round 4, GPU call 1: the fused level-1 + last kernel against the round-3 pair, first parity check on the GPU
O=$GRAFT_REPO_ROOT/gpurun_out/r4_1; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-subrecords > $O/lap_final1.json 2> $O/err1.txt
LVM_LAP_FINAL1=0 timeout 600 python bench.py --no-subrecords > $O/lap_unfused.json 2> $O/err2.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-subrecords > $O/lap_final1_driver_shape.json 2> $O/err3.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_schedules.py -q -x -k "laplace or rccl or graph" > $O/pytest_laplace.txt 2>&1
tail -5 $O/pytest_laplace.txt
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4_1"
for f in ("lap_final1","lap_unfused","lap_final1_driver_shape"):
    try:
        d=json.load(open(O+"/"+f+".json"))
        print(f, d["value"], d["verified"], d["verification"]["float_rel_err_probe"] if d["verification"] else None)
        for k,v in d["kernels"].items(): print("   ",k, v["avg_us"], v.get("gbs"))
    except Exception as e: print(f, "ERR", e)
PY
