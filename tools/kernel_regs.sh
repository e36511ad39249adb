#!/bin/bash
# registers / spills / LDS of the kernels in a compiled object of the product build: tools/kernel_regs.sh a2s_dec_persist [name-filter]
OBJ=/root/repo/piano_a2s_amd/csrc/_obj/$1.o
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=/tmp/$1.fatbin $OBJ
T=$($B/clang-offload-bundler --list --type=o --input=/tmp/$1.fatbin | grep gfx950)
$B/clang-offload-bundler --type=o --targets=$T --input=/tmp/$1.fatbin --output=/tmp/$1.co --unbundle
$B/llvm-readelf --notes /tmp/$1.co | grep -E "^ +\.(name|vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):" | paste - - - - - - - | sed 's/  */ /g' | grep "${2:-.}"
