#!/bin/bash
# Development loop of the lean SCP kernels: compile lean.hip with ONLY the two benchmark layouts (C2: <4,60,4,50,7,4>,
# C5: <8,60,1,50,24,0>; ~25 s instead of ~150 s for the ten shipped instantiations), as a product build and as a
# -DSRH_PROFILE build, and link each with the other objects of csrc/ (which must be up to date: `make` first) into
#   gpurun_variants/libsofacontrol_hip_dev.so / libsofacontrol_hip_devprof.so        (SRH_LIB_PATH selects one).
# Extra flags for A/B builds: tools/build_lean_dev.sh -DSOMETHING ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=$ROOT/soft-robot-control_amd/csrc
V='-DSRH_LEAN_VARIANTS(X)=X(4,60,4,50,7,4)X(8,60,1,50,24,0)'
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable"
mkdir -p $ROOT/gpurun_variants /tmp/leandev/prod /tmp/leandev/prof /tmp/include
cp $ROOT/include/*.h /tmp/include/            # common.h includes "../../include/sofacontrol_hip.h"
OTHERS=$(ls $R/*.o | grep -v '/lean.o$')
rm -f $ROOT/gpurun_variants/libsofacontrol_hip_dev.so $ROOT/gpurun_variants/libsofacontrol_hip_devprof.so     # never leave a stale library behind a failed compile
pids=()
for flavour in prod prof; do
  ( cd /tmp/leandev/$flavour && rm -f *.h *.hip && cp $R/*.h $R/lean.hip . &&
    $ROOT/tools/hipcc_guarded.sh lean.hip lean.o "$V" $([ $flavour = prof ] && echo -DSRH_PROFILE) "$@" $F &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS lean.o -o $ROOT/gpurun_variants/libsofacontrol_hip_dev$([ $flavour = prof ] && echo prof).so ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "build_lean_dev.sh: a compile failed" >&2; exit 1; }; done
ls -la $ROOT/gpurun_variants/libsofacontrol_hip_dev*.so
