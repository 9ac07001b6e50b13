#!/bin/bash
# Disassemble the gfx950 code object of a libksa build: tools/disasm.sh <lib.so> <out.s>
L=/opt/rocm/lib/llvm/bin
t=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$t/fat "$1"
tgt=$($L/clang-offload-bundler --list --type=o --input=$t/fat | grep gfx950 | head -1)
$L/clang-offload-bundler --unbundle --type=o --input=$t/fat --targets=$tgt --output=$t/co
$L/llvm-objdump -d --no-show-raw-insn $t/co | c++filt > "$2"
rm -rf $t
