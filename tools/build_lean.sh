#!/bin/bash
# Development helper: recompile only lean.hip (product build in-tree, -DSRH_PROFILE build under /tmp/prof) and relink both
# libraries; the profile library goes to gpurun_variants/libsofacontrol_hip_prof.so (SRH_LIB_PATH selects it).
# Both compiles go through tools/hipcc_guarded.sh like the Makefile's.
set -e
R=/root/repo/soft-robot-control_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable"
mkdir -p /tmp/prof/csrc /tmp/prof/sofacontrol_amd /tmp/include
cp /root/repo/include/*.h /tmp/include/
cp $R/*.h $R/*.hip /tmp/prof/csrc/
( cd $R && /root/repo/tools/hipcc_guarded.sh lean.hip lean.o $F && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 *.o -o ../sofacontrol_amd/libsofacontrol_hip.so ) &
( cd /tmp/prof/csrc && /root/repo/tools/hipcc_guarded.sh lean.hip lean.o -DSRH_PROFILE $F && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 *.o -o ../sofacontrol_amd/libsofacontrol_hip.so && cp ../sofacontrol_amd/libsofacontrol_hip.so /root/repo/gpurun_variants/libsofacontrol_hip_prof.so ) &
wait
echo built
