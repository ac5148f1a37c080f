#!/bin/bash
# Timing ablations of the patch-mode 3x3 loop (FRX_P3_ABL bits, conv_kernels.h): builds scripts/bin/libfrx_ablN.so (debug stamps)
# from the objects of the current csrc build + igemm_p3.hip recompiled per variant.  Usage: bash scripts/p3_ablate.sh build|run
set -e
R=$(cd $(dirname $0)/.. && pwd)
C=$R/face-recognition-models_amd/csrc
VARS=${VARS:-"0 1 2 4 8 16 3 19"}
if [ "$1" = build ]; then
  mkdir -p $R/scripts/bin
  for v in $VARS; do
    ( cd $C && /opt/rocm/bin/hipcc -DFRX_DBG_TIMES -DFRX_P3_ABL=$v -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-pass-failed -fno-gpu-rdc -c igemm_p3.hip -o /tmp/igemm_p3_abl$v.o &&
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/scripts/bin/libfrx_abl$v.so bn.o conv.o core.o head.o igemm_dgrad_bn.o igemm_dgrad_plain.o igemm_fwd.o misc.o verify.o wgrad.o /tmp/igemm_p3_abl$v.o ) &
  done
  wait
else
  cd $R/scripts
  for v in $VARS; do
    echo "== FRX_P3_ABL=$v"
    FRX_LIB=$R/scripts/bin/libfrx_abl$v.so timeout -k 10 100 python3 p3_stamps.py 2>&1 | grep -v amdgpu.ids
  done
fi
