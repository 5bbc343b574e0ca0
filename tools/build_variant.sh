#!/bin/bash
# Variant build of the library with extra compile flags (developer experiments; needs no GPU):
#   tools/build_variant.sh probes -DMM3DGS_PROBES     ->  mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_probes.so
# (-DMM3DGS_PROBES compiles the MM3DGS_EXP timing probes in: tools/late_probe.sh, tools/exp_ab.sh; the product build has none)
# run with   MM3DGS_LIB=$PWD/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_exactdiv.so python ...
set -e
TAG=$1; shift
SRC=$(cd "$(dirname "$0")/../mm3dgs_slam_amd/csrc" && pwd)
OBJ=/tmp/mm3dgs_variant_$TAG; mkdir -p $OBJ $SRC/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -Wall -Wno-unused-function $*"
for f in api preprocess binning composite fused loss compact; do
  /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $OBJ/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $SRC/variants/libmm3dgs_hip_$TAG.so $OBJ/*.o
ls -la $SRC/variants/libmm3dgs_hip_$TAG.so
