#!/bin/bash
# round 4, GPU call 2: A/B of build variants (LVM_HIP_LIB) and schedules on the Laplace headline
O=$GRAFT_REPO_ROOT/gpurun_out/r4_2; mkdir -p $O; cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-subrecords --steps 256 --warmup 64"
L=$GRAFT_REPO_ROOT/live-video-magnification_amd
run() { n=$1; shift; env "$@" timeout 300 $B > $O/$n.json 2> $O/$n.err; }
run base X=1
run P0 LVM_HIP_LIB=$L/liblvm_P0.so
run D1 LVM_HIP_LIB=$L/liblvm_D1.so
run D2 LVM_HIP_LIB=$L/liblvm_D2.so
run D3 LVM_HIP_LIB=$L/liblvm_D3.so
run fused1 LVM_LAP_FINAL1=1
B="$B --frames-per-call 64 --ring 128"
run T64 X=1
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4_2"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f))
        ks=" ".join("%s=%.1f"%(k,v["avg_us"]) for k,v in d["kernels"].items())
        print(os.path.basename(f), d["value"], "|", ks)
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
