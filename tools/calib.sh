#!/bin/bash
# HBM counter calibration on a GPU box: gpurun -- 'bash tools/calib.sh'.  Two --pmc passes (never combined with a trace domain
# other than the kernel trace), then the join.
O=$GRAFT_REPO_ROOT/gpurun_out/r3_calib; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/hbm_counter_calib
$B > $O/known.jsonl
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf -o t -- $B > $O/pf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw -o t -- $B > $O/pw.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/hbm_counter_calib.py $O/known.jsonl $O/pf $O/pw > $O/r03_hbm_counter_calibration.json
cat $O/r03_hbm_counter_calibration.json | head -60
