#!/bin/bash
# A/B of the streaming (nontemporal) accesses: the same sources built with -DLVM_NT=0 (liblvm_hip_nt0.so) against the default build
cd $GRAFT_REPO_ROOT
N0="LVM_HIP_LIB=$GRAFT_REPO_ROOT/live-video-magnification_amd/liblvm_hip_nt0.so"
for m in laplace riesz color; do
  echo "== $m"
  BENCH_ARGS="--mode $m" bash tools/ab.sh r5_nt_$m "$N0" "LVM_X=1" "$N0" "LVM_X=1"
done
echo "== laplace, groups of the last kernel"
bash tools/ab.sh r5_fin_groups "LVM_FIN_GROUPS=512" "LVM_FIN_GROUPS=1024" "LVM_FIN_GROUPS=2048" "LVM_FIN_GROUPS=4096" "LVM_FIN_ROWS=16" "LVM_FIN_ROWS=4"
