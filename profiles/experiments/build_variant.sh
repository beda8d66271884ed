#!/bin/bash
# usage: build_variant.sh <name> <source.hip> [-D flags...]   -> scratch/ab/lib_<name>.so
set -e
mkdir -p /root/repo/scratch/ab && cd /root/repo/scratch/ab
name=$1; srcf=$2; shift 2
C=/root/repo/denseslam-global-consistency-h_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -Wall -Wno-unused-function -I$C "$@" -c $srcf -o obj_$name.o
objs=""
for f in capi alloc raycast maintain view track mesh shard integrate; do
  case " $REPLACES " in *" $f "*) ;; *) objs="$objs $C/$f.o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib_$name.so obj_$name.o $objs
echo built lib_$name.so
