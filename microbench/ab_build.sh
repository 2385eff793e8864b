#!/bin/bash
# A/B of two builds of libjpegx.so (JPEGX_LIB_PATH) on the headline kernel, the round trip and the inverse.
# usage: microbench/ab_build.sh <other.so> ; prints interleaved A B A B results
set -e
other=$1
for rep in 1 2; do
  for lib in default "$other"; do
    if [ "$lib" = default ]; then unset JPEGX_LIB_PATH; else export JPEGX_LIB_PATH=$lib; fi
    echo "== lib=$lib rep=$rep"
    python microbench/ab_forward.py nt=0x1 --kind noise --rounds 5
    python microbench/ab_forward.py inv=0x0 skipx=0x400 --kind noise --direction inverse --rounds 5
    python microbench/ab_forward.py inv=0x0 --kind smooth --direction inverse --rounds 5
    python microbench/ab_forward.py pooled=0x1 --kind noise --pool 2 --rounds 5 --planes 4
  done
done
