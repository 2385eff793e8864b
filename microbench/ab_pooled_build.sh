#!/bin/bash
# pooled forward (and config 3 through bench.py) for two builds of libjpegx.so, A B A B
other=$1
for rep in 1 2 3; do
  for lib in default "$other"; do
    if [ "$lib" = default ]; then unset JPEGX_LIB_PATH; else export JPEGX_LIB_PATH=$lib; fi
    echo "== lib=$lib rep=$rep"
    python microbench/ab_forward.py pooled=0x1 pooled_skipx=0x401 --kind noise --pool 2 --rounds 5 --planes 4
  done
done
