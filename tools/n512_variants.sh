#!/bin/bash
# Builds the four variants of the generic kernel's translation unit that separate the two causes of round 2's wrong
# results at N > 256 and runs tools/n512_probe.py on each, on ONE box:
#   V0  compiler's own fence lowering, real function calls      (round 1 / round 2 code generation)
#   V1  explicit s_waitcnt around the agent-scope fences, real calls
#   V2  compiler's own fence lowering, every device function inlined (-mllvm -amdgpu-function-calls=false)
#   V3  both: what ships
# Usage (on the GPU box, through gpurun):  bash tools/n512_variants.sh > gpurun_out/n512_variants.txt
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
C=$R/fcc_qp_amd/csrc
T=/tmp/n512v; mkdir -p $T
COMMON="-x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -mllvm -pragma-unroll-threshold=400000 -w"
hipcc $COMMON -DFCCQP_NO_EXPLICIT_WAITS -c $C/fccqp_generic.hip -o $T/g0.o &
hipcc $COMMON -c $C/fccqp_generic.hip -o $T/g1.o &
hipcc $COMMON -DFCCQP_NO_EXPLICIT_WAITS -mllvm -amdgpu-function-calls=false -c $C/fccqp_generic.hip -o $T/g2.o &
wait
for v in 0 1 2; do
  hipcc -shared -fPIC --offload-arch=gfx950 -o $T/lib$v.so $C/fccqp_capi.o $T/g$v.o $C/fccqp_fused.o $C/fccqp_fused_f32.o $C/fccqp_fused_split.o $C/fccqp_coop.o
done
cd $R
for pass in 1 2; do
  for v in 0 1 2 3; do
    echo "=== pass $pass variant V$v"
    if [ $v -lt 3 ]; then export FCCQP_LIB_PATH=$T/lib$v.so; else unset FCCQP_LIB_PATH; fi
    N512_SKIP=2 N512_WARM=0 timeout -k 10 300 python tools/n512_probe.py 6 2>&1 | grep -v amdgpu.ids | cut -c1-120
  done
done
