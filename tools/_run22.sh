mkdir -p gpurun_out/r23
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
R="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --profile-steps 0 --steps 64 --warmup 32"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/gpurun_out/r23/prof -o lap_p1 -- $R > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $GRAFT_REPO_ROOT/gpurun_out/r23/prof -o lap_p2 -- $R > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/r23/prof -o lap_p3 -- $R > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/r23/prof
