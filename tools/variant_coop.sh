#!/bin/bash
# Builds tools/ab/libfccqp_hip_<tag>.so: the cooperative kernel's translation units (fccqp_coop.hip, fccqp_coop_split.hip)
# recompiled with extra flags (e.g. a -D experiment switch), linked with the objects of the last full build -- for same-box
# A/B runs (FCCQP_LIB_PATH=tools/ab/libfccqp_hip_<tag>.so python tools/coop_split_ab.py ...).
# Usage: bash tools/variant_coop.sh <tag> [extra hipcc flags...]
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/fcc_qp_amd/csrc
O=/tmp/fccqp_variant_$TAG; mkdir -p $O $ROOT/tools/ab
FLAGS="-x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -falign-loops=256 -mllvm -pragma-unroll-threshold=400000 -Wno-unused-function -Wno-unused-variable"
for f in fccqp_coop fccqp_coop_split; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $C/$f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/tools/ab/libfccqp_hip_$TAG.so $C/fccqp_capi.o $C/fccqp_jit.o $C/fccqp_generic.o $C/fccqp_fused.o $C/fccqp_fused_f32.o $C/fccqp_fused_split.o $C/fccqp_check.o $O/fccqp_coop.o $O/fccqp_coop_split.o -ldl
ls -la $ROOT/tools/ab/libfccqp_hip_$TAG.so
