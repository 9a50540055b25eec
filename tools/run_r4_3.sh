#!/bin/bash
# round 4, GPU call 3: the per-frame (live) surface -- lvm_process_device, one frame per call
O=$GRAFT_REPO_ROOT/gpurun_out/r4_3; mkdir -p $O; cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-subrecords --steps 256 --warmup 64 --frames-per-call 1"
run() { n=$1; shift; env "$@" timeout 300 $B $EXTRA > $O/$n.json 2> $O/$n.err; }
run base X=1
run d0fused LVM_D0_FUSED_WAVES=1
run d0fused_r8 LVM_D0_FUSED_WAVES=1 LVM_D0_MIN_TASKS=100000
run fusedown3 LVM_FUSE_DOWN=3
run nosplit LVM_LAP_SPLIT_MIN_NT=4
EXTRA="--pipeline 1" run pipe1 X=1
EXTRA="--graph" run graph X=1
EXTRA="--streams 4" run B4 X=1
EXTRA="--streams 4" run B4_d0fused LVM_D0_FUSED_WAVES=1
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4_3"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f))
        ks=" ".join("%s=%.1f"%(k,v["avg_us"]) for k,v in d["kernels"].items())
        print(os.path.basename(f), d["value"], "us/frame %.1f"%(1e3*d["ms_per_step"]), "host %.1f"%(1e3*d["host_enqueue_ms_per_step"]), "|", ks)
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
