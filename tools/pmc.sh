#!/bin/bash
# rocprofv3 passes over one bench command (GPU box).  Usage: tools/pmc.sh OUTDIR "bench args" ["COUNTERS pass 1" "COUNTERS pass 2" ...]
# Pass 0 is always a plain --kernel-trace --stats run; counter passes never combine with other trace domains.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; ARGS="$2"; shift 2
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-subrecords --no-clock-probe --profile-steps 0 $ARGS"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p0 -o t -- $B > $OUT/p0.log 2>&1
i=0
for c in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p$i -o t -- $B > $OUT/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
dirs=""; for j in $(seq 0 $i); do dirs="$dirs $OUT/p$j"; done
python tools/pmc_fold.py $dirs > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
