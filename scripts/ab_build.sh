#!/bin/bash
# A/B variant of libshapegan_hip.so: one source recompiled with extra flags, linked with the product's other objects.
#   bash scripts/ab_build.sh r4 sdfnet.hip -DSG_BWD_RING=4     ->  scripts/_abl/r4.so   (use with SHAPEGAN_HIP_LIB=...)
name=$1; src=$2; shift; shift
root=$(cd $(dirname $0)/.. && pwd); obj=$root/shapegan_amd/csrc/_obj; out=$root/scripts/_abl; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $root/shapegan_amd/csrc/$src -o $out/$name.o || exit 1
others=$(ls $obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $out/$name.o $others && echo built $out/$name.so
