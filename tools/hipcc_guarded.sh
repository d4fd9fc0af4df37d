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
$HIPCC "$@" --cuda-device-only -S "$SRC" -o "$B.s" 2> >(grep -v 'argument unused during compilation' >&2)
python3 "$HERE/check_spill_exec.py" --fix "$B.s"
python3 "$HERE/check_dpp_hazard.py" --fix "$B.s"
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=$ARCH -c "$B.s" -o "$B.dev.o"
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$B.dev.out" "$B.dev.o"
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH \
    -input=/dev/null -input="$B.dev.out" -output="$B.hipfb"
$HIPCC "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$B.hipfb" -c "$SRC" -o "$OBJ"
rm -f "$B.dev.o" "$B.dev.out" "$B.hipfb"
