#!/bin/bash
# round 4, GPU call 4: nontemporal vs plain stores / loads in the fused first kernel (store acknowledgements sit on the path of every later load wait)
O=$GRAFT_REPO_ROOT/gpurun_out/r4_4; mkdir -p $O; cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-subrecords --steps 256 --warmup 64"
L=$GRAFT_REPO_ROOT/live-video-magnification_amd
run() { n=$1; shift; env "$@" timeout 300 $B $EXTRA > $O/$n.json 2> $O/$n.err; }
run base X=1
run S0 LVM_HIP_LIB=$L/liblvm_S0.so
run S0L0 LVM_HIP_LIB=$L/liblvm_S0L0.so
run S0D1 LVM_HIP_LIB=$L/liblvm_S0D1.so
EXTRA="--frames-per-call 1" run pf_base_fused LVM_D0_FUSED_WAVES=1
EXTRA="--frames-per-call 1" run pf_S0_fused LVM_D0_FUSED_WAVES=1 LVM_HIP_LIB=$L/liblvm_S0.so
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4_4"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f))
        ks=" ".join("%s=%.1f"%(k,v["avg_us"]) for k,v in d["kernels"].items())
        print(os.path.basename(f), d["value"], "|", ks)
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
