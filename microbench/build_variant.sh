#!/bin/bash
# Build another libjpegx.so next to the default one for A/B runs (JPEGX_LIB_PATH selects it):
#   microbench/build_variant.sh <name> <extra hipcc flags...>   ->  microbench/_ab/libjpegx_<name>.so
set -e
name=$1; shift
here=$(cd "$(dirname "$0")" && pwd)
mkdir -p $here/_ab
make -s -j8 -C $here/../implementing-jpeg-compression_amd/csrc OUT=$here/_ab/libjpegx_$name.so OBJ=_obj_$name EXTRA="$*"
ls -la $here/_ab/libjpegx_$name.so
