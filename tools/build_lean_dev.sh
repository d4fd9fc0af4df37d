#!/bin/bash
# A/B loop of the lean SCP kernels: compile lean.hip with ONLY the two benchmark layouts (C2: <4,60,4,50,7,4>, C5: <8,60,1,50,24,0>;
# ~40 s instead of minutes for all shipped instantiations) and link it with the other objects of csrc/ (which must be up to
# date: `make` first) into
#   gpurun_variants/libdev_<tag>.so                     (tools/run_ab_variants.sh / SRH_LIB_PATH select one).
# Usage: tools/build_lean_dev.sh <tag> [-DSOMETHING ...]     several tags can be built side by side (one directory each).
set -e
TAG=${1:?usage: build_lean_dev.sh <tag> [flags]}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=$ROOT/soft-robot-control_amd/csrc
V='-DSRH_LEAN_VARIANTS(X)=X(4,60,4,50,7,4)X(8,60,1,50,24,0)'
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -mllvm -disable-machine-licm"      # (the flags of csrc/Makefile for lean.o)
D=/tmp/leandev/$TAG/csrc
mkdir -p $ROOT/gpurun_variants $D /tmp/leandev/include /tmp/include
cp $ROOT/include/*.h /tmp/leandev/include/            # common.h includes "../../include/sofacontrol_hip.h"
rm -f $ROOT/gpurun_variants/libdev_$TAG.so            # never leave a stale library behind a failed compile
cd $D && rm -f *.h *.hip *.o && cp $R/*.h $R/lean.hip .
$ROOT/tools/hipcc_guarded.sh lean.hip lean.o "$V" "$@" $F
OTHERS=$(ls $R/*.o | grep -v '/lean.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS lean.o -o $ROOT/gpurun_variants/libdev_$TAG.so
grep -A40 "amdhsa_kernel.*gusto_lean_kernelILi4ELi60ELi4ELi50E" $D/lean.s | grep "private_segment_fixed_size\|next_free_vgpr" | tr -s '\t ' ' ' | tr '\n' ';'; echo " ($TAG: C2 GuSTO kernel)"
ls -la $ROOT/gpurun_variants/libdev_$TAG.so
