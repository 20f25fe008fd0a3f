#!/bin/bash
# extract_co.sh LIB.so OUT.co -- the gfx950 code object out of a built library (to diff two builds: identical md5 = identical kernels;
# disassemble with /opt/rocm/lib/llvm/bin/llvm-objdump -d OUT.co)
set -e
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 \
  --input=<(objcopy -O binary --only-section=.hip_fatbin "$1" /dev/stdout) --output="$2" --unbundle
