#!/bin/bash
# usage: dis_kernel.sh <abs path of lib.so> <kernel-name substring> <out.s>   - disassembly of ONE kernel of the gfx950 code object
set -e
mkdir -p /tmp/dis && cd /tmp/dis
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" fatbin
$L/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fatbin --output=k.co
$L/llvm-objdump -d --no-show-raw-insn k.co > k.s
n=$(grep -n "$2.*>:" k.s | head -1 | cut -d: -f1)
awk -v n=$n 'NR==n{print;next} NR>n{if ($0 ~ />:$/) exit; print}' k.s | sed 's#\s*//.*##' > "$3"
wc -l "$3"
