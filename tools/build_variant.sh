#!/bin/bash
# Experiment builds of the product library next to the shipped one: tools/build_variant.sh TAG "-DAVS_EXP_..." -> adaptiveviscositysolver_amd/exp/libavs_hip_TAG.so
# (objects in csrc/exp_TAG/; select with AVS_LIB_PATH; exp/ is git-ignored but travels to the GPU box)
set -e
TAG=$1; DEFS=$2
cd "$(dirname "$0")/../adaptiveviscositysolver_amd/csrc"
mkdir -p exp_$TAG ../exp
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -Wall -Wno-unused-function $DEFS"
for f in avs_api avs_brick avs_brick_build avs_pcg avs_assembly avs_dist avs_reorder avs_prepass avs_post; do
  ( /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o exp_$TAG/$f.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/libavs_hip_$TAG.so exp_$TAG/*.o avs_partition.o -L/opt/rocm/lib -lrccl
ls -la ../exp/libavs_hip_$TAG.so
