#!/bin/bash
# Round 4, VERDICT item 4: what the visible-list ring push costs the HBM-sized fusion launch, and in which form.
#   build (in the container):  bash profiles/experiments/push_variants.sh build
#   run (on the GPU box):      bash profiles/experiments/push_variants.sh run   -> gpurun_out/push_variants.json
# Variants of k_integrate<false,true,true> (DSLAM_PUSH_VARIANT, integrate.hip): 0 product, 1 no push, 2 ring bit only,
# 3 last_seen only, 4 load-OR-store instead of the atomic, 5 / 6 = 0 / 4 behind the group's block stores.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/denseslam-global-consistency-h_amd/csrc
AB=$R/profiles/experiments/_ab
if [ "$1" = build ]; then
  mkdir -p $AB
  for v in 0 1 2 3 4 5 6; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -Wall -Wno-unused-function -I$C -DDSLAM_PUSH_VARIANT=$v -c $C/integrate.hip -o $AB/integrate_push$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $AB/lib_push$v.so $AB/integrate_push$v.o $C/capi.o $C/alloc.o $C/raycast.o $C/maintain.o $C/view.o $C/track.o $C/mesh.o $C/shard.o
    rm -f $AB/integrate_push$v.o
  done
  ls -la $AB
else
  python $R/profiles/experiments/push_variants.py $AB/lib_push0.so $AB/lib_push1.so $AB/lib_push2.so $AB/lib_push3.so $AB/lib_push4.so $AB/lib_push5.so $AB/lib_push6.so
fi
