#!/bin/bash
# round 4, GPU call 6: the fused level-1 + last kernel with 184 / 168 VGPRs (no spills, 2 / 3 waves per SIMD)
O=$GRAFT_REPO_ROOT/gpurun_out/r4_6; mkdir -p $O; cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/live-video-magnification_amd
run() { n=$1; shift; env "$@" timeout 300 $B $EXTRA > $O/$n.json 2> $O/$n.err; }
B="python bench.py --no-cpu-baseline --no-subrecords --steps 256 --warmup 64"
run T32_base X=1
run T32_X2 LVM_HIP_LIB=$L/liblvm_X2.so LVM_LAP_FINAL1=1
run T32_X3 LVM_HIP_LIB=$L/liblvm_X3.so LVM_LAP_FINAL1=1
B="$B --frames-per-call 1"
run T1_X2 LVM_HIP_LIB=$L/liblvm_X2.so LVM_LAP_FINAL1=1
run T1_X3 LVM_HIP_LIB=$L/liblvm_X3.so LVM_LAP_FINAL1=1
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4_6"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f))
        ks=" ".join("%s=%.1f"%(k,v["avg_us"]) for k,v in d["kernels"].items())
        print(os.path.basename(f), d["value"], "us/frame %.1f"%(1e3*d["ms_per_step"]), "|", ks)
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
