mkdir -p gpurun_out/r20
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TCP_[A-Z_0-9]+|GRBM_[A-Z_0-9]+|LDS[A-Za-z_0-9]*|VALU[A-Za-z_0-9]*|MemUnit[A-Za-z]*|Occupancy[A-Za-z]*)\b" | sort -u | tr '\n' ' ' | head -c 6000 > $GRAFT_REPO_ROOT/gpurun_out/r20/counters.txt
R="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --profile-steps 0 --steps 48 --warmup 32 --mode riesz"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/gpurun_out/r20/prof -o rz_p1 -- $R > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $GRAFT_REPO_ROOT/gpurun_out/r20/prof -o rz_p2 -- $R > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/r20/prof -o rz_p3 -- $R > /dev/null 2>&1
ls -la $GRAFT_REPO_ROOT/gpurun_out/r20/prof; head -c 1500 $GRAFT_REPO_ROOT/gpurun_out/r20/counters.txt
