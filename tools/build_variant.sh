#!/bin/bash
# build a variant of the library that differs in ONE source's compile flags:
#   tools/build_variant.sh <tag> <source.hip> <extra flags...>   ->  tools/probe/bin/libasp_<tag>.so
# (same-box A / B runs: ASP_AMD_LIB=$PWD/tools/probe/bin/libasp_<tag>.so python3 bench.py ...)
set -e
TAG=$1; SRC=$2; shift 2
L=audiosignalprocess_amd/lib; C=audiosignalprocess_amd/csrc
python3 -c "from audiosignalprocess_amd.build import build_library; build_library()"
EXTRA=""
case $SRC in ns_kernels*.hip) EXTRA="-mllvm -amdgpu-kernarg-preload-count=8";; aec_kernels.hip) EXTRA="-mllvm -amdgpu-sched-strategy=iterative-ilp";; esac
mkdir -p tools/probe/bin /tmp/variant_$TAG
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function $EXTRA "$@" -Iinclude -I$C -c $C/$SRC -o /tmp/variant_$TAG/$SRC.o
OBJS=$(ls $L/*.o | grep -v "/$SRC.o")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/variant_$TAG/$SRC.o -o tools/probe/bin/libasp_$TAG.so
echo built tools/probe/bin/libasp_$TAG.so
