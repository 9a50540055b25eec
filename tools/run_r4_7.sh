#!/bin/bash
# round 4, GPU call 7: forward-table instruction diet (2 v + 1 entries, unconditional B neighbour, L cells by origin node | in 2x2x2 blocks)
O=$GRAFT_REPO_ROOT/gpurun_out/r4_7; mkdir -p $O; cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/live-video-magnification_amd
run() { n=$1; shift; env "$@" timeout 300 $B $EXTRA > $O/$n.json 2> $O/$n.err; }
B="python bench.py --no-cpu-baseline --no-subrecords --steps 256 --warmup 64"
run lap_node X=1
run lap_blocked LVM_HIP_LIB=$L/liblvm_BL.so
run lap_node2 X=1
B="$B --mode riesz"
run rz_node X=1
run rz_blocked LVM_HIP_LIB=$L/liblvm_BL.so
B="python bench.py --no-cpu-baseline --no-subrecords --steps 256 --warmup 64 --frames-per-call 1"
run pf_node X=1
run pf_blocked LVM_HIP_LIB=$L/liblvm_BL.so
python - <<'PY'
import json,os,glob
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4_7"
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.load(open(f))
        ks=" ".join("%s=%.1f"%(k,v["avg_us"]) for k,v in d["kernels"].items())
        print(os.path.basename(f), d["value"], "us/frame %.1f"%(1e3*d["ms_per_step"]), "|", ks)
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
