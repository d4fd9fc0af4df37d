#!/bin/bash
# hipcc_guarded.sh <source.hip> <object.o> [compiler flags...]
# Compiles one HIP translation unit for gfx950 the way `hipcc -c` does, but with the device assembly passed through
# tools/check_spill_exec.py --fix (see there: VGPR spills placed in front of an EXEC restore) and tools/check_dpp_hazard.py --fix
# (wait states in front of inline-asm DPP reads, which the compiler's hazard recogniser does not see) on the way:
#   device code -> assembly -> check / fix -> code object -> fat binary -> embedded by the host-side compile.
# The steps after the assembly are the ones `hipcc -###` prints for a plain `-c`.
set -e
SRC=$1; OBJ=$2; shift 2
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
LL=${LLVM_BIN:-/opt/rocm/lib/llvm/bin}
ARCH=${ARCH:-gfx950}
HERE=$(cd "$(dirname "$0")" && pwd)
B=${OBJ%.o}
# The two repair rules were written against -- and validated on -- the code generator of ONE compiler: HIP 7.2 (AMD clang 22.0.0git,
# roc-7.2.0).  Another major.minor may place spills / PHI copies / hazards differently: the rules might neither find nor fix what it
# does.  Refuse to build with it unless told that the result will be checked by other means (make check-spills + the GPU tests).
VALIDATED_HIP="7.2"
HIPV=$($HIPCC --version 2>/dev/null | sed -n 's/^HIP version: \([0-9]*\.[0-9]*\).*/\1/p' | head -n1)
if [ "$HIPV" != "$VALIDATED_HIP" ] && [ "${SRH_GUARD_UNCHECKED:-0}" != "1" ]; then
    echo "hipcc_guarded.sh: HIP version '$HIPV' != $VALIDATED_HIP, the version the assembly repair rules (check_spill_exec.py, check_dpp_hazard.py) were validated against; set SRH_GUARD_UNCHECKED=1 to build anyway" >&2
    exit 3
fi
$HIPCC "$@" --cuda-device-only -S "$SRC" -o "$B.s" 2> >(grep -v 'argument unused during compilation' >&2)
python3 "$HERE/check_spill_exec.py" --fix "$B.s"
python3 "$HERE/check_dpp_hazard.py" --fix "$B.s"
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=$ARCH -c "$B.s" -o "$B.dev.o"
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$B.dev.out" "$B.dev.o"
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH \
    -input=/dev/null -input="$B.dev.out" -output="$B.hipfb"
$HIPCC "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$B.hipfb" -c "$SRC" -o "$OBJ"
rm -f "$B.dev.o" "$B.dev.out" "$B.hipfb"
