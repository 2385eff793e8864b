#!/bin/bash
# usage: microbench/ab_libs.sh "<args for ab_forward.py>" lib1.so lib2.so ...   (default lib first; A B C A B C)
args=$1; shift
for rep in 1 2 3; do
  for lib in default "$@"; do
    if [ "$lib" = default ]; then unset JPEGX_LIB_PATH; else export JPEGX_LIB_PATH=$lib; fi
    python microbench/ab_forward.py $args | sed "s|^|$(basename $lib) rep$rep: |"
  done
done
