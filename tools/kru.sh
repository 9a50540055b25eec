#!/bin/bash
# Kernel resource usage of one translation unit of csrc/ as the Makefile builds it: name, VGPRs, SGPRs, scratch, waves per SIMD, LDS.
#   tools/kru.sh laplace.hip [extra flags]
cd "$(dirname "$0")/../live-video-magnification_amd/csrc" || exit 1
f=$1; shift
NOSLP=""; case $f in laplace.hip|riesz.hip) NOSLP=-fno-slp-vectorize;; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I../../include -I. $NOSLP "$@" \
  --cuda-device-only -c $f -o /tmp/kru.$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
 awk '/Function Name:/{n=$5} /TotalSGPRs:/{s=$4} / VGPRs:/{v=$4} /ScratchSize/{sc=$5} /Occupancy/{o=$5} /LDS Size/{print "vgpr", v, "sgpr", s, "scratch", sc, "occ", o, "lds", $6, n}' | c++filt | sed "s/(.*//"
rm -f /tmp/kru.$$.o
