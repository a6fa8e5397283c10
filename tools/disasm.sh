#!/bin/bash
# gfx950 disassembly of one object of the product build:  bash tools/disasm.sh gru  ->  /tmp/dis/gru.s
set -e
L=/opt/rocm/lib/llvm/bin
mkdir -p /tmp/dis
$L/llvm-objcopy --dump-section=.hip_fatbin=/tmp/dis/$1.fat ${2:-cpc_audio_amd/lib/obj/$1.hip.o}
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/dis/$1.fat --output=/tmp/dis/$1.co --unbundle
$L/llvm-objdump -d /tmp/dis/$1.co > /tmp/dis/$1.s
grep -n "^[0-9a-f]* <" /tmp/dis/$1.s
