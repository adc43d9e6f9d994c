#!/bin/bash
# Device assembly (gfx950) of one kernel of a .hip file:  tools/isa_of.sh csrc/batch_kernels.hip 'batch_kernelILi3ELi1ELb1ELb1ELb1E' > out.s
set -e
src=$(readlink -f "$1"); pat=$2
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip --cuda-device-only -S "$src" -o /tmp/isa_of.s 2>/dev/null
awk -v pat="$pat" '$0 ~ "^_ZN.*" pat ".*:" {f=1} f{print} /^\.Lfunc_end[0-9]*:/{if(f){exit}}' /tmp/isa_of.s
