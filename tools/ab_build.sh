#!/bin/bash
# tools/ab_build.sh VARIANT TU "FLAGS" -- an A/B build of the library under _ab/VARIANT/: the objects of the current build with ONE
# translation unit (skred_amd/csrc/TU.hip) recompiled with extra FLAGS (e.g. -DSKS_RING=8), linked into _ab/VARIANT/libskred_amd.so.
# Run here (hipcc cross-compiles); the variants travel to the GPU box with the snapshot; tools/ab_libs.sh runs a script on each.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); V=$1; TU=$2; FLAGS=$3
C=$R/skred_amd/csrc; O=$C/_obj_ab_$V
make -s -j8 -C $C ../libskred_amd.so > /dev/null
rm -rf $O; mkdir -p $O $R/_ab/$V; cp $C/_obj/*.o $O/
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -std=c++17 -I$R/include -I$C $FLAGS -c $C/$TU.hip -o $O/$TU.o
OBJS=$(ls $O/*.o | grep -v -E "skred_synth_dropin|skred_synth_persample|skred_patch")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/_ab/$V/libskred_amd.so $OBJS -ldl -lpthread
rm -rf $O
echo "built _ab/$V/libskred_amd.so ($TU with $FLAGS)"
