#!/bin/bash
# builds libshapegan_hip variants with -DSG_ABLATE=<n> into scripts/_abl/ (tuning experiments; results are wrong on purpose)
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/_abl
for n in "$@"; do
  objs=""
  for f in conv3d conv3d_halo gemm sdfnet batchnorm elementwise pointnet; do
    if [ $f = conv3d_halo ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSG_ABLATE=$n -c shapegan_amd/csrc/$f.hip -o scripts/_abl/${f}_$n.o
      objs="$objs scripts/_abl/${f}_$n.o"
    else
      objs="$objs shapegan_amd/csrc/_obj/$f.o"
    fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/_abl/lib_$n.so $objs
done
