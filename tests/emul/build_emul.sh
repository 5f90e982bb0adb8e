#!/bin/bash
# TEST INFRASTRUCTURE: compile the unmodified kernel sources for the host emulator.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${EMUL_CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT="$HERE/_build"
mkdir -p "$OUT"
SRCS=("$ROOT/deepspeaker-pytorch_amd/csrc/"*.hip)
ARGS=()
for s in "${SRCS[@]}"; do ARGS+=(-x c++ "$s"); done
"$CXX" ${EMUL_EXTRA} -std=c++17 -O2 -g -fPIC -shared -pthread -Wno-unused-function -Wno-unknown-attributes \
    -I"$HERE" -I"$ROOT/deepspeaker-pytorch_amd/csrc" -I"$ROOT/include" \
    -x c++ "$HERE/emu_runtime.cpp" "${ARGS[@]}" -o "$OUT/libdeepspeaker_emul.so"
echo "$OUT/libdeepspeaker_emul.so"
