#!/bin/bash
# Development helper: the product library in-tree (`make`: every unit through tools/hipcc_guarded.sh) and, beside it, a library
# whose lean.hip is compiled with -DSRH_PROFILE (phase clocks printed by the kernel; all instantiations) and linked with the
# in-tree objects of the other units:  gpurun_variants/libsofacontrol_hip_prof.so   (SRH_LIB_PATH selects it).
# Extra flags for the profile compile: tools/build_lean.sh -DSOMETHING ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=$ROOT/soft-robot-control_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -mllvm -disable-machine-licm"      # (the flags of csrc/Makefile for lean.o)
mkdir -p $ROOT/gpurun_variants /tmp/leanprof/csrc /tmp/include
rm -f $ROOT/gpurun_variants/libsofacontrol_hip_prof.so          # never leave a stale library behind a failed compile
cp $ROOT/include/*.h /tmp/include/                              # common.h includes "../../include/sofacontrol_hip.h"
( cd /tmp/leanprof/csrc && rm -f *.h *.hip *.o && cp $R/*.h $R/lean.hip . && $ROOT/tools/hipcc_guarded.sh lean.hip lean.o -DSRH_PROFILE "$@" $F ) &
prof=$!
make -C $R -j2
wait $prof || { echo "build_lean.sh: the profile compile failed" >&2; exit 1; }
OTHERS=$(ls $R/*.o | grep -v '/lean.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS /tmp/leanprof/csrc/lean.o -o $ROOT/gpurun_variants/libsofacontrol_hip_prof.so
ls -la $ROOT/gpurun_variants/libsofacontrol_hip_prof.so $ROOT/soft-robot-control_amd/sofacontrol_amd/libsofacontrol_hip.so
