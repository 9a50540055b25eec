#!/bin/bash
# round 4, GPU call 5: per-frame surface with the fused level-1 + last kernel (experimental build)
O=$GRAFT_REPO_ROOT/gpurun_out/r4_5; mkdir -p $O; cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-subrecords --steps 256 --warmup 64 --frames-per-call 1"
L=$GRAFT_REPO_ROOT/live-video-magnification_amd
run() { n=$1; shift; env "$@" timeout 300 $B $EXTRA > $O/$n.json 2> $O/$n.err; }
run base X=1
run X_off LVM_HIP_LIB=$L/liblvm_X.so
run X_final1 LVM_HIP_LIB=$L/liblvm_X.so LVM_LAP_FINAL1=1
EXTRA="--streams 4" run B4_base X=1
EXTRA="--streams 4" run B4_final1 LVM_HIP_LIB=$L/liblvm_X.so LVM_LAP_FINAL1=1
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4_5"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f))
        ks=" ".join("%s=%.1f"%(k,v["avg_us"]) for k,v in d["kernels"].items())
        print(os.path.basename(f), d["value"], "us/frame %.1f"%(1e3*d["ms_per_step"]), "|", ks)
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
