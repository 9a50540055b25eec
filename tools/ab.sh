#!/bin/bash
# A/B of build / schedule variants on the GPU box: prints value + the per-kernel HIP-event averages of each variant
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ab}; mkdir -p $O; shift
i=0
while [ $# -gt 0 ]; do
  v="$1"; shift; i=$((i+1))
  env $v timeout 300 python bench.py --no-cpu-baseline --no-subrecords --steps 128 --warmup 32 ${BENCH_ARGS:-} > $O/v$i.json 2> $O/v$i.err
  python - "$v" $O/v$i.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    ks=d["kernels"]
    print("%-70s fps %9.0f | "%(sys.argv[1][:70], d["value"]) + " ".join("%s %.0f"%(n.replace("pyr_down","pd").replace("lap_",""),v["avg_us"]) for n,v in ks.items()))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
