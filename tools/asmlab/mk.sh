#!/bin/bash
# mk.sh src [-D...]  : device-only assembly of the product source -> $OUT (default base.s)
# mk.sh co file.s    : assemble + link -> file.co
set -e
LLVM=/opt/rocm/lib/llvm/bin
HERE=$(cd "$(dirname "$0")" && pwd)
if [ "$1" = src ]; then
  shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 --cuda-device-only -S "$@" \
      -o ${OUT:-$HERE/base.s} $HERE/../../meta-viterbinet_amd/csrc/mvn_hip.hip 2>/dev/null
else
  f=$2
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $f -o ${f%.s}.o
  $LLVM/ld.lld -shared ${f%.s}.o -o ${f%.s}.co
  rm -f ${f%.s}.o
fi
